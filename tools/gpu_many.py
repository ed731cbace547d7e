#!/usr/bin/env python3
"""Many-object piles (BASELINE.json config 4 shape: UR5gripper_2_finger_many_objects.xml) on one MI355X: settle + step timing.
    python tools/gpu_many.py [n_scenes] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
sim = BatchSim(m, n, lib_path=os.environ.get("UR5_LIB"))
t0 = time.perf_counter()
sim.reset(20 + np.arange(n, dtype=np.uint64), 1, 1000.0)
print(f"reset + settle (491 steps): {time.perf_counter() - t0:.2f} s wall, kernel {sim.last_launch_ms():.1f} ms -> {n * 491 / sim.last_launch_ms() * 1e3:.3e} env-steps/s")
c0 = sim.counters()
sim.step(steps)
ms = sim.last_launch_ms()
c1 = sim.counters()
it = (c1["solver_iters"] - c0["solver_iters"]).sum() / (n * steps)
print(f"settled pile: {steps} steps x {n} scenes, kernel {ms:.1f} ms -> {n * steps / ms * 1e3:.3e} env-steps/s, {ms / steps * 1e3:.0f} us/step/scene-batch, "
      f"newton iters/step {it:.1f}, ncon_max {c1['ncon_max'].max()}, status {np.bitwise_or.reduce(c1['status'])}")
