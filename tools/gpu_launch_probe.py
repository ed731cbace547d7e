#!/usr/bin/env python3
"""Residency probe: kernel time (HIP events) of the reset + 1000 ms settle launch (500 physics steps per scene, PID holding the arm) for a
sweep of batch sizes. UR5_LIB=<.so> selects the build."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
m = load_model("it1_4box")
for n in [int(x) for x in sys.argv[1:]] or [2, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]:
    sim = BatchSim(m, n, lib_path=os.environ.get('UR5_LIB'))
    ms = []
    for rep in range(2):
        sim.reset(np.arange(n, dtype=np.uint64) + 20, 1, 1000.0)
        ms.append(sim.last_launch_ms())
    steps = 491
    print("n=%5d  settle %.2f / %.2f ms  -> %.1f us per step per wave-slot, %.3f M env-steps/s" % (n, ms[0], ms[1], ms[1] * 1e3 / steps, n * steps / ms[1] / 1e3), flush=True)
    sim.close()
