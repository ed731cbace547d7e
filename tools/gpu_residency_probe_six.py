#!/usr/bin/env python3
"""How many pile scenes does the chip hold at once? The profile build stamps every scene's start / end (s_memrealtime); a settle launch of n scenes shows how many
workgroups started at once = resident slots. Variants: the handle's private stream, torch's default stream, a torch side stream (what bench.py uses).
    UR5_PROF_LIB=tools/libur5sim_prof.so python tools/gpu_residency_probe.py [n=1024]"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = load_model("/UR5+gripper/UR5gripper_2_finger.xml")
lib = os.environ.get("UR5_PROF_LIB", "tools/libur5sim_prof.so")


def probe(tag, stream=None):
    sim = BatchSim(m, n, lib_path=lib)
    sim.lib.ur5_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    if stream is not None:
        sim.set_stream(stream)
    sim.reset(np.arange(n, dtype=np.uint64) + 20, 1, 400.0)
    out = np.zeros((n, 26))
    sim.lib.ur5_profile_read(sim._h, out.ctypes.data_as(C.POINTER(C.c_double)))
    t0 = out[:, 16].min()
    st, en = (out[:, 16] - t0) / 1e5, (out[:, 17] - t0) / 1e5
    steps = sim.counters()["total_steps"].astype(float)
    print(f"{tag:28s} n {n}: started within 1 ms: {int((st < 1.0).sum())}, kernel {sim.last_launch_ms():.0f} ms, scene lifetime median {np.median(en - st):.0f} ms, "
          f"us/step median {np.median((en - st) * 1e3 / steps):.0f}, max concurrently alive {int(max(((st <= t) & (en > t)).sum() for t in np.linspace(0, en.max(), 200)))}")
    sim.close()


import torch  # noqa: E402
torch.zeros(1, device="cuda")
probe("private stream (torch loaded)")
probe("torch default stream", torch.cuda.current_stream().cuda_stream)
s = torch.cuda.Stream()
probe("torch side stream", s.cuda_stream)
