#!/usr/bin/env python3
"""Render throughput: 200x200 RGB-D of N scenes, device buffers, HIP-event time."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = load_model("it1_4box")
sim = BatchSim(m, n)
sim.reset(np.arange(n, dtype=np.uint64) + 20, 1, 200.0)
img = torch.zeros((n, 200, 200, 3), dtype=torch.uint8, device="cuda"); dep = torch.zeros((n, 200, 200), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
ts = []
for i in range(5):
    sim.render_dev(img.data_ptr(), dep.data_ptr(), 1, 200, 200, 0); sim.sync(); ts.append(sim.last_launch_ms())
ms = float(np.median(ts[1:]))
byts = n * 200 * 200 * 7
print("render %d scenes 200x200: %.2f ms -> %.0f frames/s, %.1f GB/s written (algorithmic 280 kB/frame), %.4f of 8 TB/s" % (n, ms, n / ms * 1e3, byts / ms / 1e6, byts / ms / 1e6 / 8000))
