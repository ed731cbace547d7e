#!/usr/bin/env python3
"""How far from the oracle do the OBJECTS of a small scene end after one aimed attempt (check_mode 0: the grasped box is carried over the drop bin and released)?
Per scene: reward / step counts equal?, arm error, largest error of an untouched box, error of the aimed box.   python tools/gpu_drop_parity.py <lib> [n=24]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import aimed_actions
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from oracle.oracle import Oracle
lib, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
m = load_model("it1_4box")
sim = BatchSim(m, n, lib_path=lib)
seeds = 20 + np.arange(n, dtype=np.uint64)
sim.reset(seeds, 1, 1000.0)
st = sim.get_state()
acts = aimed_actions(st["qpos"], 4)
rots = (np.arange(n) // 4) % 6
rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0)
s2 = sim.get_state()["qpos"]
arm, others, aimed, same = [], [], [], 0
for e in range(n):
    o = Oracle(m)
    o.reset(int(seeds[e]), 1, True)
    r, pso, pro = o.grasp_attempt(acts[e], int(rots[e]), 0)
    q = o.get_state()["qpos"]
    same += int(r == rew[e] and pso.tolist() == ps[e].tolist())
    eo = np.abs(s2[e][8:] - q[8:]).reshape(-1, 7)[:, :3].max(axis=1)
    arm.append(np.abs(s2[e][:8] - q[:8]).max()); others.append(np.delete(eo, e % 4).max()); aimed.append(eo[e % 4])
print(f"{os.path.basename(lib):26s} n {n}: reward+steps equal {same}/{n}, successes {int(rew.sum())}; arm max {max(arm):.1e}; untouched boxes max {max(others):.1e}; "
      f"aimed box median {np.median(aimed):.1e} max {max(aimed):.1e}; kernel {sim.last_launch_ms():.0f} ms")
