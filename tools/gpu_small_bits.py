#!/usr/bin/env python3
"""Two builds of the wavefront-per-scene kernels must produce the SAME BITS (a change that only moves data, e.g. the six-object Hessian staged in two row panels):
n scenes, reset + 1000 ms settle + `rounds` rounds (observation + aiming rule + attempt + episode resets inside the launch) through lib A and lib B; state records,
rewards, action records and the rendered depth frames compared exactly.     python tools/gpu_small_bits.py libA libB [it4|it1] [n] [rounds]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
la, lb = sys.argv[1], sys.argv[2]
kind = sys.argv[3] if len(sys.argv) > 3 else "it4"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 512
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 5
m = load_model("/UR5+gripper/UR5gripper_2_finger.xml" if kind == "it4" else "it1_4box")
dev = torch.device("cuda", 0)
res = []
for lib in (la, lb):
    sim = BatchSim(m, n, lib_path=lib)
    sim.reset((20 + np.arange(n)).astype(np.uint64), 1, 1000.0)
    sim.set_stream(torch.cuda.current_stream().cuda_stream)
    wl = bench.It1Rounds(torch, m, sim, dev, 0, n, n, "aimed", kind)
    rew = torch.zeros((rounds, n), dtype=torch.int32, device=dev)
    act, px = wl.launch_rounds(0, rounds, rew)
    sim.sync()
    c = sim.counters()
    out = [sim.state_tensor("cuda").clone().cpu().numpy().view(np.uint64), rew.cpu().numpy(), act.cpu().numpy().view(np.uint64), c["total_steps"], c["solver_iters"]]
    if kind != "it1":
        out.append(wl._frames[1].cpu().numpy().view(np.uint32))
    res.append(out)
    ms = sim.last_launch_ms()
    del wl, sim
names = ["state records", "rewards", "action records", "step counters", "Newton iterations", "depth frames"]
same = [bool(np.array_equal(a, b)) for a, b in zip(*res)]
for k, s in zip(names, same):
    print("%-20s %s" % (k, "identical" if s else "DIFFERENT"))
print("%s: %d scenes x %d rounds, %.0f env-steps per scene, success %.3f, last launch %.1f ms: %s"
      % (kind, n, rounds, res[0][3].mean(), res[0][1].mean(), ms, "BIT-IDENTICAL" if all(same) else "NOT identical"))
sys.exit(0 if all(same) else 1)
