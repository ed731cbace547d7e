#!/usr/bin/env python3
"""Why a grasped object is lost on the way to the drop position: a per-contact trace of one scripted grasp on the CPU oracle.

One object of the pile scene alone on the bin floor (tools/shape_grasp_table.py's models), the reference's move_and_grasp sequence
(GraspingEnv.py:205-386) issued call by call so that the state can be read in between. After the gripper has closed, every `--every` steps
of the lift and of the carry it prints, in the ee_link frame: the object's position, and per object contact the distance, the normal force
(sum of the 10 pyramid rows of a condim-6 contact), sliding / torsional / rolling friction, the relative tangential velocity the solver sees
(read back from the reference accelerations: aref+ - aref- = -2 b mu v_t) and the contact normal.

    python tools/diag_carry_slip.py --shape 1 --pose 0 --rot 0 --z floor      # sphere r 0.025, fingertips at floor level + 2 cm

What it shows (profiles/r03_carry_slip_sphere.txt): the grip itself is static -- held still after the lift the sphere does not creep
(|v| = 0, 2.4 s) -- but during the 2.2 m/s swing to the drop position (Kp[0] = 10, GraspingEnv.py:282) all three contacts slide along the pads'
width at ~40 mm/s while the friction rows carry only ~0.3 N each, far inside their cones (N = 1300 N). That is the constraint model working
as specified [3P, MuJoCo computation chapter]: constraints act on J qacc, the velocity-dependent term Jdot qvel is dropped, and the reference
acceleration aref = -b J qvel (b = 2 / (dmax * timeconst) = 202 1/s for solref .01 1) removes 40 % of the relative velocity per 2 ms step. A
gripper swinging at omega = 3.1 rad/s around the base re-creates omega^2 r = 5.8 m/s^2 (+ tangential) of relative acceleration every step, so
the sliding speed settles at a_c / b = 40 mm/s -- whatever the normal force. Pads are 22 mm wide: a point contact (sphere, cylinder held
across) leaves them in ~0.3 s, a box merely shifts. The grasp bit of such attempts is decided by kinematics of the carry, not by friction.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import shape_grasp_table as T  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, default=1)
ap.add_argument("--pose", type=int, default=0)
ap.add_argument("--rot", type=int, default=0)
ap.add_argument("--z", default="top", help="'top': the depth-derived height (object's top surface); 'floor': the bin floor (fingertips stop at TABLE_HEIGHT)")
ap.add_argument("--every", type=int, default=40)
ap.add_argument("--hold", type=int, default=0, help="instead of carrying: hold still for this many ms after the lift")
a = ap.parse_args()
np.set_printoptions(linewidth=220, precision=3, suppress=True)
m = T.model_for(a.shape)
sh = T.SHAPES[a.shape]
print(f"{sh[0]} | {sh[3][a.pose][0]} | rotation index {a.rot} | z from {a.z}")
o = Oracle(m)
o.reset(20, 1, False)
st = o.get_state()
st["qpos"][8:15], _ = T.case_state(None, m, sh[3][a.pose])
o.set_state(qpos=st["qpos"], qvel=np.zeros(m.nv))
o.stay(1000)
q = o.get_state()["qpos"]
x, y = q[8], q[9]
top = q[10] + sh[3][a.pose][3] if a.z == "top" else T.FLOOR
eeb, gb, objb = m.body_name2id("ee_link"), m.geom_bodyid, m.nbody - 1
B = 2.0 / (0.99 * 0.01)


def frame(n):   # oracle make_frame()
    yv = np.array([0, 1, 0.]) if abs(n[1]) < 0.5 else np.array([0, 0, 1.])
    t1 = yv - n * np.dot(n, yv)
    t1 /= np.linalg.norm(t1)
    return n, t1, np.cross(n, t1)


def show(tag):
    s = o.get_state()
    qq, c, r, R, p = s["qpos"], o.contacts(), o.rows(), o.body_xmat()[eeb], o.body_xpos()[eeb]
    k = len(r) - sum(1 if int(cc[9]) == 1 else 2 * (int(cc[9]) - 1) for cc in c)
    print(f"{tag:18s} grip {qq[6]:+.3f} object in ee frame {np.round(R.T @ (qq[8:11] - p), 4)} |v| {np.linalg.norm(s['qvel'][8:11]):.2f} m/s, body rate {np.round(s['qvel'][11:14], 2)}")
    for cc in c:
        nr = 1 if int(cc[9]) == 1 else 2 * (int(cc[9]) - 1)
        f, ar = r[k:k + nr, 4], r[k:k + nr, 1]
        k += nr
        if gb[int(cc[7])] != objb and gb[int(cc[8])] != objb:
            continue
        n, t1, t2 = frame(cc[4:7])
        mu = [1.0, 1.0, 0.8, 0.8, 0.8]
        vt = (-(ar[0] - ar[1]) * t1 - (ar[2] - ar[3]) * t2) / (2 * B)
        ft = (f[0] - f[1]) * t1 + (f[2] - f[3]) * t2
        other = m.names["body"][gb[int(cc[8])] if gb[int(cc[7])] == objb else gb[int(cc[7])]]
        print(f"    {other[:20]:20s} dist {cc[0] * 1e3:6.3f} mm  N {f.sum():7.1f}  sliding friction (ee) {np.round(R.T @ ft, 1)} N  v_t (ee) {np.round(R.T @ vt * 1e3, 1)} mm/s"
              f"  torsion {mu[2] * (f[4] - f[5]):6.2f}  rolling {mu[3] * (f[6] - f[7]):6.2f} {mu[4] * (f[8] - f[9]):6.2f}  normal (ee) {np.round(R.T @ n, 3)}  active rows {(f > 0).sum()}")


print("above target", o.move_ee([x, y, 1.1], 0.05, 1000), "rotate", o.move_group(1 << 5, np.array([np.deg2rad([0, 30, 60, 90, -30, -60][a.rot])]), 0.05, 500),
      "open", o.open_gripper(True), "descend", o.move_ee([x, y, max(0.91, top - 0.01)], 0.01, 300))
o.stay(100)
print("grasp(): close_gripper(300) ->", o.close_gripper(300), "(1 = max. steps reached = something is between the fingers)")
show("closed")
n = 0
while o.move_ee([0, -0.6, 1.1], 0.05, a.every)[0] != 0 and n < 100:
    n += 1
show("lifted")
if a.hold:
    for k in range(a.hold // 400):
        o.stay(400)
        show(f"held {400 * (k + 1)} ms")
    sys.exit(0)
n = 0
while True:
    r = o.move_ee([0.6, 0.0, 1.15], 0.01, a.every)
    n += 1
    show(f"carry step {n * a.every}")
    if r[0] == 0 or n > 60:
        break
print("closing check: close_gripper(1000) ->", o.close_gripper(1000), "(1 = still holding = reward 1)")
show("after the check")
