#!/usr/bin/env python3
"""Kernel time of a whole settle (1000 ms = 500 steps, dense piles at the end) and of a grasp round for n piles: the tree's library against an A/B library.
    python tools/gpu_residency_probe2.py lib n"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from pile_aim import pick_box
lib, n = sys.argv[1], int(sys.argv[2])
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
sim = BatchSim(m, n, lib_path=lib)
sim.reset(np.arange(n, dtype=np.uint64) + 20, 1, 1000.0)
t_settle = sim.last_launch_ms()
st = sim.get_state()
acts, rots = np.zeros((n, 3)), np.arange(n) % 6
acts[:] = [0.0, -0.6, 1.0]
for e in range(n):
    b = pick_box(m, st["qpos"][e])
    if b is not None:
        acts[e], rots[e] = b[1], b[2]
c0 = sim.counters()["total_steps"].sum()
rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0)
t_grasp = sim.last_launch_ms()
steps = sim.counters()["total_steps"].sum() - c0
print(f"{os.path.basename(lib):28s} n {n:5d}: settle kernel {t_settle:8.1f} ms ({n * 491 / t_settle:7.1f} k env-steps/s)   grasp kernel {t_grasp:8.1f} ms, {steps / n:6.0f} steps/scene "
      f"({steps / t_grasp:7.1f} k env-steps/s), longest scene {ps.sum(1).max()} steps, success {rew.mean():.3f}, status {int(sim.counters()['status'].max())}")
