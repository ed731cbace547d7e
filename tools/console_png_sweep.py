#!/usr/bin/env python3
"""media/console.png of the reference records ONE step of an older version of the script (no rotation, table top at 0.89): pixel (136, 80)
-> world (-0.1655, -0.5080, 0.89), then 362 / 136 / 202 / 631 / 33 physics steps for "move to pre grasp", "move to grasping position",
"move to center", "move to drop position", "open gripper". This sweep replays that sequence on the CPU oracle for the model / controller
variants the recording leaves open and prints which one reproduces it; the result is recorded in DESIGN.md and
tests/golden/console_png.json ("replay").

    python tools/console_png_sweep.py [/root/reference]
"""
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.mjcf import compile_mjcf  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
TARGET = np.array([-0.16551974, -0.50804459, 0.88999999])
RECORDED = [362, 136, 202, 631, 33]


def replay(o, p_scale, start, drop_tol=0.01):
    st = o.get_state()
    pid = st["pid"].copy()
    pid[:, 3] = np.array([7, 10, 5, 7, 5, 5, 2.5]) * p_scale          # MujocoController.py:160-235, "p_scale = 1" / "p_scale = 3"
    o.set_state(pid=pid)
    if start == "drop":                                                # the step before ended at the drop position, gripper open
        o.move_ee([0.6, 0.0, 1.15], 0.01, 1200)
        o.open_gripper(False)
    steps = []
    r, n = o.move_ee([TARGET[0], TARGET[1], 1.1], 0.05, 1000); steps.append(n)      # "Above target"
    o.open_gripper(True)
    r, n = o.move_ee([TARGET[0], TARGET[1], 0.91], 0.01, 300); steps.append(n)   # the recording's table top was at 0.89, today's is at 0.91
    o.stay(100)
    o.close_gripper(300)
    r, n = o.move_ee([0.0, -0.6, 1.1], 0.05, 1000); steps.append(n)
    r, n = o.move_ee([0.6, 0.0, 1.15], drop_tol, 1200); steps.append(n)
    o.close_gripper(1000)
    r = o.open_gripper(False); steps.append(o.last_steps)
    return steps


rows = []
for mi, p_scale, start, drop_tol in itertools.product(("dedup", "signed", "legacy"), (1, 3), ("home", "drop"), (0.01, 0.05)):
    m = compile_mjcf(os.path.join(REF, "UR5+gripper", "UR5gripper_2_finger.xml"), mesh_inertia=mi)
    o = Oracle(m)
    o.reset(20, 1, True)
    s = replay(o, p_scale, start, drop_tol)
    err = float(np.mean(np.abs(np.log(np.array(s, dtype=float) / np.array(RECORDED)))))
    rows.append(dict(mesh_inertia=mi, p_scale=p_scale, start=start, drop_tol=drop_tol, steps=[int(x) for x in s], mean_abs_log_ratio=round(err, 3)))
    print(rows[-1], flush=True)
best = min(rows, key=lambda r: r["mean_abs_log_ratio"])
print("recorded", RECORDED, "best", best)
json.dump(dict(recorded=RECORDED, sweep=rows, best=best), open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r02_console_png_sweep.json"), "w"), indent=1)
