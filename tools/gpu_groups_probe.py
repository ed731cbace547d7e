#!/usr/bin/env python3
"""Does splitting the 4096-scene batch into G scene groups (one handle + one HIP stream each, rounds issued back to back) keep the GPU's
wave slots full across round boundaries? Same workload as bench.py (It1Rounds), wall-clock env-steps/s for G in argv (default 1 2 4)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
N, WARM, K = 4096, 4, 8
m = load_model("it1_4box")
dev = torch.device("cuda", 0)
STAGGER = float(os.environ.get("UR5_STAGGER", "0"))      # seconds between the first launches of consecutive groups (0: back to back)
for G in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
    n = N // G
    sims, wls, streams, rews = [], [], [], []
    for g in range(G):
        s = BatchSim(m, n, device_id=0)
        s.reset(bench.BASE_SEED + np.arange(g * n, (g + 1) * n, dtype=np.uint64), 1, 1000.0)
        st = torch.cuda.Stream()
        s.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            wls.append(bench.It1Rounds(torch, m, s, dev, g * n, n, N, "aimed"))
            rews.append(torch.zeros((WARM + K, n), dtype=torch.int32, device=dev))
        sims.append(s); streams.append(st)
    def rounds(r0, r1, stagger=0.0):
        for r in range(r0, r1):
            for g in range(G):
                if stagger and r == r0 and g > 0:
                    time.sleep(stagger)
                with torch.cuda.stream(streams[g]):
                    wls[g].launch(r, rews[g][r])
    rounds(0, WARM, STAGGER)
    torch.cuda.synchronize()
    c0 = sum(int(s.counters()["total_steps"].sum()) for s in sims)
    t0 = time.perf_counter()
    rounds(WARM, WARM + K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = sum(int(s.counters()["total_steps"].sum()) for s in sims)
    succ = sum(float(r[WARM:].float().mean()) for r in rews) / G
    print("stagger %.2f s, groups %d x %d scenes: %.3f M env-steps/s, %.1f ms per round of all %d scenes, success %.3f, steps/attempt %.1f" % (
        STAGGER, G, n, (c1 - c0) / dt / 1e6, dt / K * 1e3, N, succ, (c1 - c0) / (K * N)), flush=True)
    for s in sims: s.close()
