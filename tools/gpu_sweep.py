#!/usr/bin/env python3
"""Throughput vs number of resident scenes (reset + settle launch only): shows how waves per CU overlap."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
m = load_model("it1_4box")
for n in [int(x) for x in sys.argv[1:]] or [256, 1024, 1280, 1536, 1792, 2048, 4096]:
    sim = BatchSim(m, n, lib_path=os.environ.get('UR5_LIB'))
    sim.reset(np.arange(n, dtype=np.uint64) + 20, 1, 1000.0)
    ms = sim.last_launch_ms(); steps = sim.counters()["total_steps"].sum()
    print("n=%5d (%.2f/CU): settle kernel %.1f ms, %.3e env-steps/s" % (n, n / 256, ms, steps / ms * 1e3), flush=True)
    sim.close()
