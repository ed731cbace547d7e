#!/usr/bin/env python3
"""BASELINE.json config 4 shape on one MI355X: N many-object piles (UR5gripper_2_finger_many_objects.xml, the reference's default
GraspEnv), rendered 200x200 RGB-D observation, multi-discrete [pixel, rotation] actions aimed at an object, full grasp script.
    python tools/gpu_many_grasp.py [n_scenes] [rounds] [many|it4]
it4 = BASELINE.json config 3 shape: the in-tree UR5gripper_2_finger.xml (3 boxes + 3 spheres), same rendered observation + depth-based z."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_rl_ur5_amd.envs import GraspEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.perf_counter()
which = sys.argv[3] if len(sys.argv) > 3 else "many"
env = GraspEnv(n_envs=n, show_obs=False, observation="render") if which == "many" else \
    GraspEnv(file="/UR5+gripper/UR5gripper_2_finger.xml", n_envs=n, show_obs=False, observation="render")
obs = env.reset()
print(f"create + reset (drop, settle 1000 ms, render): {time.perf_counter() - t0:.2f} s; depth range {obs['depth'].min():.3f} .. {obs['depth'].max():.3f} m")
c0 = env.sim.counters()
for r in range(rounds):
    xp = env.sim.body_xpos()[:, 8:8 + (env.model.nv - 8) // 6]            # world positions of the objects
    acts = np.zeros((n, 2), dtype=np.int64)
    for e in range(n):
        objs = xp[e]
        inbin = np.where((np.abs(objs[:, 0]) < 0.2) & (np.abs(objs[:, 1] + 0.6) < 0.13) & (objs[:, 2] > 0.85))[0]
        k = inbin[(e + r) % len(inbin)] if len(inbin) else 0
        px, py = env.controller.world_2_pixel(objs[k, :3])
        acts[e] = [int(np.clip(py, 0, 199)) * 200 + int(np.clip(px, 0, 199)), (e + r) % 6]
    t0 = time.perf_counter()
    obs, reward, done, info = env.step(acts)
    dt = time.perf_counter() - t0
    c1 = env.sim.counters()
    steps = (c1["total_steps"] - c0["total_steps"]).sum()
    c0 = c1
    print(f"round {r}: {dt:.2f} s wall, {steps} env-steps -> {steps / dt:.3e} env-steps/s, {n / dt:.1f} grasp attempts/s, success {reward.mean():.2f}, "
          f"skipped {info['skipped'].mean():.2f}, status {np.bitwise_or.reduce(c1['status'])}, ncon_max {c1['ncon_max'].max()}, "
          f"phase steps mean {np.asarray(info['phase_steps']).mean(axis=0).astype(int).tolist()}")
