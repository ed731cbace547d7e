#!/usr/bin/env python3
"""Quick GPU sanity run (used through gpurun): HIP engine vs the CPU oracle on a few scenes + a first timing."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from oracle.oracle import Oracle

m = load_model("it1_4box")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nref = 4
t0 = time.time(); sim = BatchSim(m, n); print("create %.2fs" % (time.time() - t0), flush=True)
seeds = np.arange(n, dtype=np.uint64) + 20
t0 = time.time(); sim.reset(seeds, 1, 1000.0); dt = time.time() - t0
c = sim.counters()
print("reset+settle: %.3fs wall, kernel %.3f ms, steps/env %s, env-steps/s %.3e" % (dt, sim.last_launch_ms(), c["total_steps"][:4], c["total_steps"].sum() / (sim.last_launch_ms() * 1e-3)), flush=True)
st = sim.get_state()
ors = []
for e in range(nref):
    o = Oracle(m); o.reset(int(seeds[e]), 1, True); ors.append(o)
    so = o.get_state()
    print("settle env", e, "qpos err %.3e" % np.abs(st["qpos"][e] - so["qpos"]).max(), "steps", o.total_steps, c["total_steps"][e])
acts = np.zeros((n, 3)); rots = np.zeros(n, dtype=int)
for e in range(n):
    objs = st["qpos"][e][8:].reshape(-1, 7); k = e % 4
    acts[e] = [objs[k, 0], -0.6 + objs[k, 1], 0.91]; rots[e] = (e // 4) % 6
t0 = time.time(); rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0); dt = time.time() - t0
c2 = sim.counters()
steps = (c2["total_steps"] - c["total_steps"])
print("grasp: wall %.3fs kernel %.3f ms, mean steps %.0f, env-steps/s %.3e, success rate %.3f, status %s, newton it/step %.2f" % (
    dt, sim.last_launch_ms(), steps.mean(), steps.sum() / (sim.last_launch_ms() * 1e-3), rew.mean(), np.unique(c2["status"]),
    (c2["solver_iters"] - c["solver_iters"]).sum() / steps.sum()), flush=True)
s2 = sim.get_state()
for e in range(nref):
    r, pso, pro = ors[e].grasp_attempt(acts[e], int(rots[e]), 0)
    so = ors[e].get_state()
    print("grasp env", e, "reward", rew[e], r, "phase steps equal", bool((pso == ps[e]).all()), "arm err %.3e" % np.abs(s2["qpos"][e][:8] - so["qpos"][:8]).max(),
          "qpos err %.3e" % np.abs(s2["qpos"][e] - so["qpos"]).max())
