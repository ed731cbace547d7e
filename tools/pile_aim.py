"""Aiming rule for 40-object piles (tools/gpu_many_agreement.py, tests/test_many_objects.py): the box a 1 cm-deep top grasp can hold.

What this scene's physics holds with the reference's grip depth (GraspingEnv.py:258-259: fingertips 1 cm below the surface the depth image reports) was
measured on the oracle with single objects (tools/shape_grasp_table.py): boxes whose sides are parallel to the fingers -- wrist angle = -yaw within ~10
degrees (mod 90) --, upright cylinders at 90 degrees, little else. The rule aims at the box with the most level top face, the least yaw misalignment to
one of the wrist angles 0 / 30 / -30 degrees (rotation indices 0 / 1 / 4 of GraspingEnv.py:40) and nothing lying on it, at the height of its top face.
"""
import numpy as np


def quat_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def pick_box(m, qpos):
    """(object index, xyz of the grasp action, rotation index, score) of the best box of one scene (qpos of the whole scene), or None.
    score = tilt of the top face + yaw misalignment in degrees (+100 when another object lies on the box); below ~12 a grasp is plausible."""
    best = None
    nobj = (m.nq - 8) // 7
    P = qpos[8:].reshape(-1, 7)
    geom0 = m.ngeom - nobj
    for k in range(nobj):
        if m.geom_type[geom0 + k] != 6:
            continue
        c, R, half = P[k, :3], quat_mat(P[k, 3:7]), m.geom_size[geom0 + k]
        if not (abs(c[0]) < 0.17 and abs(c[1] + 0.6) < 0.10 and c[2] > 0.89):
            continue                                                            # inside the bin, away from its walls
        a = int(np.argmax(np.abs(R[2])))
        tilt = np.degrees(np.arccos(min(1.0, abs(R[2, a]))))
        b = (a + 1) % 3
        yaw = np.degrees(np.arctan2(R[1, b], R[0, b]))
        want = -yaw                                                             # fingers parallel to the box's sides (oracle probe: yaw 30 <-> wrist -30)
        cand = {0: 0.0, 1: 30.0, 4: -30.0}
        mis = {r: abs(((want - ang + 45) % 90) - 45) for r, ang in cand.items()}
        r = min(mis, key=mis.get)
        d = P[:, :3] - c
        on_top = np.any((np.hypot(d[:, 0], d[:, 1]) < 0.05) & (d[:, 2] > 0.01) & (np.arange(nobj) != k))
        score = tilt + mis[r] + (100.0 if on_top else 0.0)
        if best is None or score < best[3]:
            best = (k, [c[0], c[1], c[2] + half[a] * abs(R[2, a])], r, score)
    return best
