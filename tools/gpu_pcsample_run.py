#!/usr/bin/env python3
"""Workload for `rocprofv3 --pc-sampling-beta-enabled`: reset + settle, then ONE aimed grasp-attempt launch of n scenes (default 2048)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
m = load_model("it1_4box")
sim = BatchSim(m, n)
sim.reset(np.arange(n, dtype=np.uint64) + 20, 1, 1000.0)
st = sim.get_state(); acts = np.zeros((n, 3))
for e in range(n):
    objs = st["qpos"][e][8:].reshape(-1, 7); k = e % 4
    acts[e] = [objs[k, 0], -0.6 + objs[k, 1], 0.91]
rew, ps, pr = sim.grasp_attempt(acts, rot=0, check_mode=1)
print("grasp launch %.1f ms, success %.3f" % (sim.last_launch_ms(), rew.mean()))
