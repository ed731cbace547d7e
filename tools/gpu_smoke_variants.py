#!/usr/bin/env python3
"""Runs one small launch of each engine kernel in its own process and reports which survive (debugging aid for codegen faults)."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SNIP = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
m = load_model(%r)
sim = BatchSim(m, %d)
sim.reset(np.arange(%d, dtype=np.uint64) + 20, 1, 0.0)
op = %r
if op == "step": sim.step(5)
elif op == "settle": sim.stay(1000.0)
elif op == "grasp":
    sim.stay(1000.0)
    print(sim.grasp_attempt(np.array([0.0, -0.6, 0.95]), rot=0, check_mode=1)[0][:4])
elif op == "forward": sim.forward_debug()
print("OK", sim.counters()["total_steps"][:2], sim.last_launch_ms())
'''
for model, n in (("it1_4box", 8), ("/UR5+gripper/UR5gripper_2_finger.xml", 4), ("/UR5+gripper/UR5gripper_2_finger_many_objects.xml", 2)):
    for op in ("step", "forward", "settle", "grasp"):
        r = subprocess.run([sys.executable, "-c", SNIP % (ROOT, model, n, n, op)], capture_output=True, text=True, timeout=300)
        tail = (r.stdout.strip().splitlines() or [""])[-1] if r.returncode == 0 else (r.stderr.strip().splitlines() or ["?"])[0][:150]
        print(f"{model[-28:]:28s} {op:8s} rc={r.returncode} {tail}", flush=True)
