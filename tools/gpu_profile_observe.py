#!/usr/bin/env python3
"""Cycles of the in-launch observation (Engine::observe) per round and scene, profile build (make -C mujoco_rl_ur5_amd/csrc prof): one launch of `rounds` rounds of the rendered
workload with the observation rendered and the rule evaluated inside the launch; x4 of the profile = observe cycles.    python tools/gpu_profile_observe.py [many|it4] [n] [rounds]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
kind = sys.argv[1] if len(sys.argv) > 1 else "many"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml" if kind == "many" else "/UR5+gripper/UR5gripper_2_finger.xml")
sim = BatchSim(m, n, lib_path=os.environ.get("UR5_PROF_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libur5sim_prof.so")))
sim.lib.ur5_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
sim.reset((20 + np.arange(n)).astype(np.uint64), 1, 1000.0)
dev = torch.device("cuda", 0)
sim.set_stream(torch.cuda.current_stream().cuda_stream)
wl = bench.It1Rounds(torch, m, sim, dev, 0, n, n, "aimed", kind)
rew = torch.zeros((rounds, n), dtype=torch.int32, device=dev)
c0 = sim.counters()["total_steps"].astype(float)
wl.launch_rounds(0, rounds, rew)
sim.sync()
out = np.zeros((n, 26)); sim.lib.ur5_profile_read(sim._h, out.ctypes.data_as(C.POINTER(C.c_double)))
steps = sim.counters()["total_steps"].astype(float) - c0
print("%s: %d scenes x %d rounds in one launch, kernel %.1f ms, %.0f steps per scene; observe: %.0f cycles per round and scene (%.3f ms at 2.4 GHz) = %.2f %% of the scene's cycles in steps (sum of phases %.3e per scene)"
      % (kind, n, rounds, sim.last_launch_ms(), steps.mean(), out[:, 22].mean() / rounds, out[:, 22].mean() / rounds / 2.4e6, 100 * out[:, 22].sum() / out[:, :16].sum(), out[:, :16].sum(1).mean()))
