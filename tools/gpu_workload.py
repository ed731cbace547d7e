#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 runs: settle + one grasp round on N scenes (default 1024)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = load_model("it1_4box")
sim = BatchSim(m, n, lib_path=os.environ.get('UR5_LIB'))
sim.reset(np.arange(n, dtype=np.uint64) + 20, 1, 1000.0)
st = sim.get_state()["qpos"]
acts = np.zeros((n, 3))
for e in range(n):
    o = st[e][8:].reshape(-1, 7); k = e % 4
    acts[e] = [o[k, 0], -0.6 + o[k, 1], 0.91]
c0 = sim.counters()["total_steps"].sum()
rew, ps, pr = sim.grasp_attempt(acts, rot=np.arange(n) % 6, check_mode=1)
c1 = sim.counters()["total_steps"].sum()
c = sim.counters()
print("ncon_max histogram:", np.bincount(c["ncon_max"])[10:].tolist(), "status!=0:", int((c["status"] != 0).sum()))
for r in range(int(os.environ.get("EXTRA_ROUNDS", "0"))):
    st = sim.get_state()["qpos"]
    for e in range(n):
        o = st[e][8:].reshape(-1, 7); k = (e + r + 1) % 4
        acts[e] = [o[k, 0], -0.6 + o[k, 1], 0.91]
    sim.grasp_attempt(acts, rot=np.arange(n) % 6, check_mode=1)
    c = sim.counters(); print("round", r + 2, "ncon_max histogram from 10:", np.bincount(c["ncon_max"])[10:].tolist(), "status!=0:", int((c["status"] != 0).sum()))
print("grasp kernel %.1f ms, %d env-steps, %.3e env-steps/s, success %.2f" % (sim.last_launch_ms(), c1 - c0, (c1 - c0) / sim.last_launch_ms() * 1e3, rew.mean()))
