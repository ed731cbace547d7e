#!/usr/bin/env python3
"""The reference's example_agent.py (random agent on the default GraspEnv) on this engine -- BASELINE.json configs[0] as a plumbing
check on the GPU. Only the import / make line differs from the reference's script; N_ENVS > 1 steps that many scenes per call.

    python examples/example_agent.py [n_envs] [episodes] [steps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_rl_ur5_amd.envs import make

N_ENVS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N_EPISODES = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N_STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10

env = make("gym_grasper:Grasper-v0", show_obs=False, render=False, n_envs=N_ENVS)
env.print_info()

for episode in range(1, N_EPISODES + 1):
    obs = env.reset()
    for step in range(N_STEPS):
        t0 = time.perf_counter()
        action = env.action_space.sample() if N_ENVS == 1 else np.stack([env.action_space.sample() for _ in range(N_ENVS)])
        observation, reward, done, _ = env.step(action, record_grasps=True)
        print("EPISODE {} STEP {}: reward {} ({:.2f} s), rgb {} depth {}".format(
            episode, step + 1, reward if N_ENVS == 1 else float(np.mean(reward)), time.perf_counter() - t0,
            np.asarray(observation["rgb"]).shape, np.asarray(observation["depth"]).shape))

env.close()
print("Finished.")
