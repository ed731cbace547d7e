#!/usr/bin/env python3
"""Grasp-outcome agreement on 40-object piles (BASELINE configs[3]): HIP many-object kernel vs the fp64 CPU oracle, both started from the
kernel's settled state, one full move_and_grasp script per scene. Oracle scenes run on host threads (ctypes releases the GIL).
Prints one JSON line; run through gpurun, keep the result under profiles/.   python tools/gpu_many_agreement.py [n=128] [threads=128]"""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from oracle.oracle import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(n, os.cpu_count() or 8)
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
sim = BatchSim(m, n)
sim.reset(7000 + np.arange(n, dtype=np.uint64), 1, 1000.0)
st, ctrl = sim.get_state(), sim.get_ctrl()
xpos = sim.body_xpos()[:, 8:48]
acts, rots = np.zeros((n, 3)), np.arange(n) % 6
for e in range(n):
    inbin = np.where((np.abs(xpos[e][:, 0]) < 0.2) & (np.abs(xpos[e][:, 1] + 0.6) < 0.13) & (xpos[e][:, 2] > 0.85))[0]
    k = inbin[e % len(inbin)]
    acts[e] = [xpos[e][k, 0], xpos[e][k, 1], xpos[e][k, 2] + 0.02]
rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0)
kernel_ms = sim.last_launch_ms()
s2 = sim.get_state()["qpos"]


def one(e):
    o = Oracle(m)
    o.set_state(qpos=st["qpos"][e], qvel=st["qvel"][e], warmstart=st["warmstart"][e], pid=st["pid"][e])
    o.set_ctrl(ctrl[e])
    r, pso, pro = o.grasp_attempt(acts[e], int(rots[e]), 0)
    return r, pso, pro, o.get_state()["qpos"]


t0 = time.time()
with ThreadPoolExecutor(max_workers=threads) as ex:
    res = list(ex.map(one, range(n)))
bits = sum(int(r == rew[e]) for e, (r, _, _, _) in enumerate(res))
codes = sum(int(pro.tolist() == pr[e].tolist()) for e, (_, _, pro, _) in enumerate(res))
steps = sum(int(pso.tolist() == ps[e].tolist()) for e, (_, pso, _, _) in enumerate(res))
arm = [float(np.abs(s2[e][:8] - q[:8]).max()) for e, (_, _, _, q) in enumerate(res)]
print(json.dumps(dict(scenes=n, grasp_bit_agreement=bits / n, phase_result_codes_identical=codes / n, phase_steps_identical=steps / n,
                      arm_abs_error_median=float(np.median(arm)), arm_abs_error_max=float(np.max(arm)),
                      success_rate_gpu=float(rew.mean()), success_rate_oracle=float(np.mean([r for r, _, _, _ in res])),
                      status_nonzero=int((sim.counters()["status"] != 0).sum()), kernel_ms=kernel_ms, oracle_seconds=round(time.time() - t0, 1),
                      oracle_threads=threads, note="both sides start from the kernel's settled state; piles are chaotic, so identical "
                      "step counts are not expected for every scene")))
