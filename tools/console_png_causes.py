#!/usr/bin/env python3
"""media/console.png once more (round-3 verdict, item 3): the replay of its five phases runs the two CARRIES (back above the table centre: 202 recorded steps;
to the drop position: 631) about twice as fast as the recording while the first two phases and the gripper are within 15-20 %. This sweep tests the candidate
causes one at a time on the oracle and records what each does to the five counts:
  * the shoulder-pan gain retune Kp[0] = 10 that the in-tree script applies before the carries (GraspingEnv.py:282) -- the plain replay leaves it at 21,
  * where the arm starts (home pose / the drop position the previous step ended at),
  * a wall-clock PID time step (simple_pid uses time.monotonic(): the derivative term scales with h / dt_real),
  * the tolerances of the two carry moves and the controller's p_scale (MujocoController.py:160-162 keeps "p_scale = 1" as a comment).
Result (profiles/r04_console_png_causes.json): none of them, alone or combined, brings BOTH carries within 15 % while keeping the first two phases --
p_scale = 1 fits carry 1 (0.98) but then phase 2 takes 1.6 x and the drop move never converges (1201 steps); Kp[0] = 10 slows carry 2 by 20 %, a wall-clock dt
of 0.5 ms by another 20 % (0.61 x the recording together). The recording is of an older script (table top 0.89 m, no wrist rotation); what else differed
cannot be read off one screenshot, so tests/test_oracle_kat.py keeps +-15 % on phases 1, 2, +-20 % on the gripper and a factor-2 bound on the carries.
    python tools/console_png_causes.py > profiles/r04_console_png_causes.json"""
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

m = load_model("/UR5+gripper/UR5gripper_2_finger.xml")
g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "console_png.json")))
x, y = g["world"][0], g["world"][1]
REC = np.array(g["phase_steps"], dtype=float)


def run(kp0_carry=None, pid_dt=0.0, start="home", tol3=0.05, tol4=0.01, p_scale=None):
    o = Oracle(m)
    if pid_dt:
        o.set_options(1, pid_dt, 0)
    o.reset(20, 1, True)
    if p_scale is not None:
        pid = o.get_state()["pid"].copy()
        pid[:, 3] = np.array([7, 10, 5, 7, 5, 5, 2.5]) * p_scale
        o.set_state(pid=pid)
    if start == "drop":
        o.move_ee([0.6, 0.0, 1.15], 0.01, 1200)
        o.open_gripper(False)
    steps = []
    steps.append(o.move_ee([x, y, 1.1], 0.05, 1000)[1])
    o.open_gripper(True)
    steps.append(o.move_ee([x, y, 0.91], 0.01, 300)[1])
    o.stay(100)
    o.close_gripper(300)
    if kp0_carry is not None:
        pid = o.get_state()["pid"].copy()
        pid[0, 3] = kp0_carry
        o.set_state(pid=pid)
    steps.append(o.move_ee([0.0, -0.6, 1.1], tol3, 1000)[1])
    steps.append(o.move_ee([0.6, 0.0, 1.15], tol4, 1200)[1])
    o.close_gripper(1000)
    o.open_gripper(False)
    steps.append(o.last_steps)
    return steps


rows = []
for kp0, dt, start in itertools.product((None, 10.0), (0.0, 0.0005, 0.001, 0.004), ("home", "drop")):
    s = run(kp0, dt, start)
    rows.append(dict(kp0_carry=kp0, pid_dt=dt, start=start, tol3=0.05, tol4=0.01, p_scale=3, steps=s, ratio=np.round(np.array(s) / REC, 2).tolist()))
for tol3, tol4, kp0, ps in ((0.01, 0.01, None, None), (0.01, 0.002, 10.0, None), (0.05, 0.01, None, 1), (0.05, 0.01, None, 1.5), (0.05, 0.01, None, 2), (0.05, 0.01, 10.0, 2)):
    s = run(kp0, 0.0, "home", tol3, tol4, ps)
    rows.append(dict(kp0_carry=kp0, pid_dt=0.0, start="home", tol3=tol3, tol4=tol4, p_scale=ps or 3, steps=s, ratio=np.round(np.array(s) / REC, 2).tolist()))
ok = [r for r in rows if all(abs(v - 1) <= 0.15 for v in r["ratio"][:4])]
print(json.dumps(dict(recorded=REC.tolist(), variants=rows, variants_with_the_first_four_phases_within_15_percent=len(ok),
                      best_by_mean_abs_log_ratio=min(rows, key=lambda r: float(np.mean(np.abs(np.log(np.array(r["ratio"])))))),
                      conclusion="no tested cause (Kp[0] retune, start pose, wall-clock PID dt, carry tolerances, p_scale) reproduces both carries together with phases 1-2; "
                                 "the recording's script version differs in ways one screenshot does not show"), indent=1))
