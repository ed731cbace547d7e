#!/usr/bin/env python3
"""Does the batched config-5 loop LEARN, and at which cadence?  (round-5 verdict, next-round item 5.)

The reference's trainer (Grasping_Agent_multidiscrete.py:515-579) decays epsilon over its run, takes one optimiser step per transition and resets the scene every
STEPS_PER_EPISODE steps. This runs `rounds` rounds of mujoco_rl_ur5_amd.agent.BatchedGraspAgent.round() on `n` scenes (episodes of `episode` rounds, then reset_model), with
`updates` optimiser steps per round and a replay ring of `mem` transitions, and records per round: success of all / greedy / random actions, loss, epsilon. One JSON line.
    python tools/gpu_dqn_learning.py [it1|many] [n=512] [rounds=60] [updates=64] [mem=2000] [episode=4|10] [save.pt]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_rl_ur5_amd.agent import BatchedGraspAgent

which = sys.argv[1] if len(sys.argv) > 1 else "it1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 60
updates = int(sys.argv[4]) if len(sys.argv) > 4 else 64
mem = int(sys.argv[5]) if len(sys.argv) > 5 else 2000
episode = int(sys.argv[6]) if len(sys.argv) > 6 else (4 if which == "it1" else 10)
save = sys.argv[7] if len(sys.argv) > 7 else None
kw = dict(file="it1_4box", check_mode=1) if which == "it1" else {}
agent = BatchedGraspAgent(n_envs=n, device="cuda", mem_size=mem, max_updates_per_round=updates, pipeline_groups=2, **kw)
per_round, t0 = [], time.perf_counter()
for r in range(rounds):
    if r % episode == 0:
        for e in agent.envs:
            e.reset()
    out = agent.round()
    rew, gr = out["reward"].float(), out["greedy"]
    per_round.append(dict(round=r, epsilon=round(out["epsilon"], 4), success=float(rew.mean()), greedy_share=float(gr.float().mean()),
                          greedy_success=float(rew[gr].mean()) if bool(gr.any()) else None, random_success=float(rew[~gr].mean()) if bool((~gr).any()) else None,
                          loss=(sum(out["losses"]) / len(out["losses"])) if out["losses"] else None, optimiser_steps=len(out["losses"])))
torch.cuda.synchronize()
wall = time.perf_counter() - t0
if save:
    agent.save(save)
mean = lambda xs: (sum(xs) / len(xs)) if xs else None
first, last = per_round[:10], per_round[-10:]
rnd = mean([p["random_success"] for p in per_round if p["random_success"] is not None])
summary = dict(scene=which, scenes=n, rounds=rounds, rounds_per_episode=episode, max_updates_per_round=updates, mem_size=mem, optimiser_steps=agent.learner.updates_done,
               update_to_data=agent.learner.updates_done / (n * rounds), wall_s=round(wall, 1),
               success_first_10_rounds=mean([p["success"] for p in first]), success_last_10_rounds=mean([p["success"] for p in last]),
               greedy_success_first_10_rounds=mean([p["greedy_success"] for p in first if p["greedy_success"] is not None]),
               greedy_success_last_10_rounds=mean([p["greedy_success"] for p in last if p["greedy_success"] is not None]),
               random_action_success_all_rounds=rnd, loss_first_10=mean([p["loss"] for p in first if p["loss"] is not None]), loss_last_10=mean([p["loss"] for p in last if p["loss"] is not None]),
               greedy_rotations=dict(agent.greedy_rotations), greedy_rotations_successes=dict(agent.greedy_rotations_successes))
summary["greedy_last_10_over_random"] = (summary["greedy_success_last_10_rounds"] / rnd) if (rnd and summary["greedy_success_last_10_rounds"] is not None) else None
print(json.dumps(dict(summary=summary, per_round=per_round)))
