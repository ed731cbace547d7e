#!/usr/bin/env python3
"""Two builds of the pile kernel must produce the SAME BITS (a change that only moves work, e.g. the forward substitution riding along with the factorisation):
n piles, reset + 1000 ms settle + one aimed grasp attempt each through lib A and lib B; states, rewards and phase step counts compared exactly.
    python tools/gpu_many_bits.py libA libB [n]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from pile_aim import pick_box
la, lb = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
res = []
for lib in (la, lb):
    sim = BatchSim(m, n, lib_path=lib)
    sim.reset(np.arange(n, dtype=np.uint64) + 20, 1, 1000.0)
    st = sim.get_state()
    acts, rots = np.zeros((n, 3)), np.arange(n) % 6
    acts[:] = [0.0, -0.6, 1.0]
    for e in range(n):
        b = pick_box(m, st["qpos"][e])
        if b is not None:
            acts[e], rots[e] = b[1], b[2]
    rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0)
    s2 = sim.get_state()
    res.append((st["qpos"], st["qvel"], s2["qpos"], s2["qvel"], s2["warmstart"], rew, ps, pr, sim.counters()["solver_iters"]))
    del sim
names = ["settled qpos", "settled qvel", "qpos after the attempt", "qvel", "warm start", "rewards", "phase steps", "phase results", "Newton iterations"]
same = [bool(np.array_equal(a, b)) for a, b in zip(*res)]
for k, s in zip(names, same):
    print("%-24s %s" % (k, "identical" if s else "DIFFERENT"))
print("%d piles, %d env-steps each on average, success %.3f: %s" % (n, res[0][6].sum() / n, res[0][5].mean(), "BIT-IDENTICAL" if all(same) else "NOT identical"))
sys.exit(0 if all(same) else 1)
