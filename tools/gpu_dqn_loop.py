#!/usr/bin/env python3
"""BASELINE.json config 5 shape on one MI355X: many-object piles + pixel-wise grasp-Q CNN inference per scene and round + one
learning step per round, everything device-resident (mujoco_rl_ur5_amd/agent.py).
    python tools/gpu_dqn_loop.py [n_scenes] [rounds] [it1|many]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_rl_ur5_amd.agent import BatchedGraspAgent
from mujoco_rl_ur5_amd.envs import GraspEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
which = sys.argv[3] if len(sys.argv) > 3 else "many"
env = GraspEnv(n_envs=n, show_obs=False, observation="render") if which == "many" else GraspEnv(file="it1_4box", n_envs=n, show_obs=False, observation="render", check_mode=1)
env.reset()
agent = BatchedGraspAgent(env=env, device="cuda")
c0 = env.sim.counters()["total_steps"].sum()
for r in range(rounds):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    obs = env.observation_device("cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    state = agent.transform_observation(obs)
    action, greedy = agent.epsilon_greedy(state, obs); torch.cuda.synchronize(); t2 = time.perf_counter()
    reward, skipped = env.step_device(agent.transform_action(action), obs["depth"], "cuda"); t3 = time.perf_counter()
    agent.memory.push(state, action, reward)
    loss = agent.learn(); torch.cuda.synchronize(); t4 = time.perf_counter()
    c1 = env.sim.counters()["total_steps"].sum()
    print(f"round {r}: render {1e3 * (t1 - t0):.1f} ms | transform + CNN forward ({n} x 4x200x200) + action {1e3 * (t2 - t1):.1f} ms | grasp kernel {1e3 * (t3 - t2):.1f} ms "
          f"({c1 - c0} env-steps) | replay push + learn {1e3 * (t4 - t3):.1f} ms | total {t4 - t0:.2f} s -> {n / (t4 - t0):.1f} grasp attempts/s, "
          f"{(c1 - c0) / (t4 - t0):.3e} env-steps/s; reward {float(reward.float().mean()):.3f} skipped {float(skipped.float().mean()):.2f} eps {agent.eps_threshold:.3f} loss {loss}")
    c0 = c1
