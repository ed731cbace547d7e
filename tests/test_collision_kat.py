"""Closed-form known answers for the collision routines that oracle and engine share as algorithm text (box-box SAT + clipping, MPR,
capsule-capsule): each case has an analytic contact set that neither implementation is consulted for. Run against
the oracle, the lane-emulation build and -- under -m gpu -- the HIP kernels (the cooperative 8-lanes-per-pair MPR of the small engine is
only exercised there and by the grasp parity tests).

Scene: the reference's 40-object file (spheres, boxes, cylinders, capsules on free joints); every object is parked far away on the ground
plane except the ones a case places. The pick-bin plate is a static box whose top is at z = 0.89 (UR5gripper_2_finger_many_objects.xml:120).
"""
import numpy as np
import pytest

from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from oracle.oracle import Oracle

MANY = "/UR5+gripper/UR5gripper_2_finger_many_objects.xml"
TOP = 0.89
BOX, CYL, CAP = 6, 5, 3


@pytest.fixture(scope="module")
def model():
    return load_model(MANY)


def _parked(model):
    o = Oracle(model)
    o.reset(20, 1, False)
    q = o.get_state()["qpos"].copy()
    gb = np.asarray(model.geom_bodyid)
    adr = {}
    for gi in range(model.ngeom):
        b = int(gb[gi])
        if model.body_jntnum[b] == 1 and model.jnt_type[model.body_jntadr[b]] == 0:
            adr[gi] = int(model.jnt_qposadr[model.body_jntadr[b]])
    for k, (gi, a) in enumerate(sorted(adr.items())):
        q[a:a + 7] = [3.0 + 0.4 * (k % 8), 2.0 + 0.4 * (k // 8), 0.2, 1, 0, 0, 0]
    return q, adr


def _geoms(model, gtype, size=None):
    out = []
    for g in range(model.ngeom):
        if int(model.geom_type[g]) == gtype and model.body_treeid[int(model.geom_bodyid[g])] > 0:
            if size is None or np.allclose(model.geom_size[g][:len(size)], size):
                out.append(g)
    return out


def _quat(axis, angle):
    axis = np.asarray(axis, dtype=float) / np.linalg.norm(axis)
    return [np.cos(angle / 2), *(np.sin(angle / 2) * axis)]


def _qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return [w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2]


def _contacts_oracle(model, q, geoms):
    o = Oracle(model)
    o.set_state(qpos=q, qvel=np.zeros(model.nv))
    o.forward()
    return [(c[0], c[1:4], c[4:7]) for c in o.contacts() if int(c[7]) in geoms or int(c[8]) in geoms]


def _contacts_engine(model, q, geoms, lib_path):
    sim = BatchSim(model, 1, lib_path=lib_path)
    sim.set_state(qpos=q[None], qvel=np.zeros((1, model.nv)))
    d = sim.forward_debug()
    assert sim.counters()["status"][0] == 0
    # the dump carries the engine's own geom numbering: select by place instead (everything else is parked >= 2 m away from the pick bin)
    return [(c[0], c[1:4], c[4:7]) for c in d["contacts"][0][:int(d["ncon"][0])] if abs(c[1]) < 0.5 and abs(c[2] + 0.6) < 0.5]


def _backends(model, q, geoms, emul_lib, gpu):
    yield "oracle", _contacts_oracle(model, q, geoms)
    if gpu:
        yield "hip", _contacts_engine(model, q, geoms, None)
    else:
        yield "emulation", _contacts_engine(model, q, geoms, emul_lib)


# ------------------------------------------------------------------------------------------------ the cases (closed-form expectations)
def case_box_face_on_plate(model):
    """A 4 cm cube pressed d = 0.3 mm into the plate, turned 30 deg about z: its 4 bottom corners, each at depth -d, normal +-z, contact
    point half-way between the two surfaces."""
    q, adr = _parked(model)
    g = _geoms(model, BOX, [0.02, 0.02, 0.02])[0]
    d, ang = 3e-4, np.deg2rad(30)
    q[adr[g]:adr[g] + 7] = [0.03, -0.58, TOP + 0.02 - d, *_quat([0, 0, 1], ang)]
    c, s = np.cos(ang), np.sin(ang)
    corners = [[0.03 + 0.02 * (c * sx - s * sy), -0.58 + 0.02 * (s * sx + c * sy), TOP - 0.5 * d] for sx in (-1, 1) for sy in (-1, 1)]
    return q, {g}, dict(dists=[-d] * 4, points=corners, normal_axis=[0, 0, 1])


def case_box_edge_edge(model):
    """Two 5 cm cubes, the lower one rolled 45 deg about x (top edge along x at height sqrt(2) a above its centre), the upper one rolled
    45 deg about y (bottom edge along y): the edges cross at right angles, gap = dz - 2 sqrt(2) a. One contact at the crossing point."""
    q, adr = _parked(model)
    ga, gb = _geoms(model, BOX, [0.025, 0.025, 0.025])[:2]
    a, pen = 0.025, 4e-4
    z0 = 2.0                                                                  # in mid-air: nothing else nearby
    dz = 2 * np.sqrt(2) * a - pen
    q[adr[ga]:adr[ga] + 7] = [0.0, -0.6, z0, *_quat([1, 0, 0], np.pi / 4)]
    q[adr[gb]:adr[gb] + 7] = [0.0, -0.6, z0 + dz, *_quat([0, 1, 0], np.pi / 4)]
    return q, {ga, gb}, dict(dists=[-pen], points=[[0.0, -0.6, z0 + np.sqrt(2) * a - 0.5 * pen]], normal_axis=[0, 0, 1])


def case_cylinder_upright_on_plate(model):
    """A cylinder standing on the plate (general convex pair -> MPR): one contact, depth -d along z, somewhere under its base disc."""
    q, adr = _parked(model)
    g = _geoms(model, CYL)[0]
    r, h = float(model.geom_size[g][0]), float(model.geom_size[g][1])
    d = 5e-4
    q[adr[g]:adr[g] + 7] = [0.02, -0.61, TOP + h - d, 1, 0, 0, 0]
    return q, {g}, dict(dists=[-d], normal_axis=[0, 0, 1], within_disc=([0.02, -0.61], r), z=TOP - 0.5 * d, tol=5e-6)


def case_cylinder_tilted_on_plate(model):
    """The same cylinder tilted 25 deg about x: the lowest point of its rim is h cos(t) + r sin(t) below the centre; MPR must find that
    depth (to its 1e-6 tolerance) and the plate's normal. The contact POINT is not asserted: MPR (like libccd's ccdMPRPenetration, which the
    reference runs through MuJoCo [3P]) reads it off the last portal, whose three plate-side support points are corners of the 56 cm plate,
    so it only lands somewhere under the cylinder."""
    q, adr = _parked(model)
    g = _geoms(model, CYL)[0]
    r, h = float(model.geom_size[g][0]), float(model.geom_size[g][1])
    t, d = np.deg2rad(25), 4e-4
    reach = h * np.cos(t) + r * np.sin(t)
    q[adr[g]:adr[g] + 7] = [0.02, -0.61, TOP + reach - d, *_quat([1, 0, 0], t)]
    return q, {g}, dict(dists=[-d], normal_axis=[0, 0, 1], tol=2e-5, within_disc=([0.02, -0.61], h + r), z=TOP - 0.5 * d)


def case_capsules_crossed(model):
    """Two capsules in mid-air, axes along x and along y, centres dz apart: one contact, dist = dz - r1 - r2, on the common normal."""
    q, adr = _parked(model)
    g1, g2 = _geoms(model, CAP)[:2]
    r1, r2 = float(model.geom_size[g1][0]), float(model.geom_size[g2][0])
    pen, z0, s = 2e-4, 2.0, np.sqrt(0.5)
    dz = r1 + r2 - pen
    q[adr[g1]:adr[g1] + 7] = [0.0, -0.6, z0, s, 0, s, 0]                    # axis (local z) -> world x
    q[adr[g2]:adr[g2] + 7] = [0.0, -0.6, z0 + dz, s, s, 0, 0]               # axis -> world -y
    return q, {g1, g2}, dict(dists=[-pen], points=[[0.0, -0.6, z0 + r1 - 0.5 * pen]], normal_axis=[0, 0, 1])


def case_capsules_parallel(model):
    """Parallel capsules side by side, shifted 1 cm along their axes: two contacts at the ends of the overlap of the two segments."""
    q, adr = _parked(model)
    g1, g2 = _geoms(model, CAP)[:2]
    r1, r2 = float(model.geom_size[g1][0]), float(model.geom_size[g2][0])
    h1, h2 = float(model.geom_size[g1][1]), float(model.geom_size[g2][1])
    pen, z0, s, sh = 3e-4, 2.0, np.sqrt(0.5), 0.01
    dy = r1 + r2 - pen
    q[adr[g1]:adr[g1] + 7] = [0.0, -0.6, z0, s, 0, s, 0]
    q[adr[g2]:adr[g2] + 7] = [sh, -0.6 + dy, z0, s, 0, s, 0]
    lo, hi = max(-h1, sh - h2), min(h1, sh + h2)
    pts = [[x, -0.6 + r1 - 0.5 * pen, z0] for x in (lo, hi)]
    return q, {g1, g2}, dict(dists=[-pen, -pen], points=pts, normal_axis=[0, 1, 0])


def _case_flush_cubes(model, shift, turn):
    q, adr = _parked(model)
    ga, gb = _geoms(model, BOX, [0.025, 0.025, 0.025])[:2]
    a, pen, z0 = 0.025, 3e-4, 2.0
    q[adr[ga]:adr[ga] + 7] = [0.0, -0.6, z0, 1, 0, 0, 0]
    q[adr[gb]:adr[gb] + 7] = [shift, -0.6, z0 + 2 * a - pen, *_quat([0, 0, 1], turn)]
    pts = [[x, -0.6 + y, z0 + a - 0.5 * pen] for x in (shift - a if shift > 0 else -a, a) for y in (-a, a)]
    return q, {ga, gb}, dict(dists=[-pen] * 4, points=pts, normal_axis=[0, 0, 1])


def case_equal_cubes_stacked_flush(model):
    """Two equal cubes, one exactly on top of the other (what the reference's own qpos0 holds for its objects): every incident corner lies ON
    two reference edge lines. Four contacts at the corners of the common face, whichever way those ties round."""
    return _case_flush_cubes(model, 0.0, 0.0)


def case_equal_cubes_flush_shifted(model):
    """The upper cube shifted 1 cm along x: two of its edges run ALONG reference edge lines. The overlap rectangle's four corners."""
    return _case_flush_cubes(model, 0.01, 0.0)


def case_equal_cubes_flush_quarter_turn(model):
    """The upper cube turned 90 deg about z (cos = 6e-17): ties that are off by rounding, not exact."""
    return _case_flush_cubes(model, 0.0, np.pi / 2)


CASES = [case_equal_cubes_stacked_flush, case_equal_cubes_flush_shifted, case_equal_cubes_flush_quarter_turn, case_box_face_on_plate, case_box_edge_edge, case_cylinder_upright_on_plate, case_cylinder_tilted_on_plate, case_capsules_crossed,
         case_capsules_parallel]


def _check(name, backend, contacts, exp):
    tol, ptol = exp.get("tol", 1e-9), exp.get("ptol", 1e-6)
    assert len(contacts) == len(exp["dists"]), (name, backend, len(contacts))
    ax = np.asarray(exp["normal_axis"], dtype=float)
    for dist, pos, nrm in contacts:
        assert abs(abs(np.dot(nrm, ax)) - 1) < max(1e-9, tol), (name, backend, nrm)      # along the expected axis (sign = geom order)
        assert min(abs(dist - e) for e in exp["dists"]) < tol, (name, backend, dist, exp["dists"])
    if "points" in exp:
        for p in exp["points"]:
            assert min(np.abs(np.asarray(pos) - p).max() for _, pos, _ in contacts) < ptol, (name, backend, p, [c[1] for c in contacts])
    if "within_disc" in exp:
        (cx, cy), r = exp["within_disc"]
        for _, pos, _ in contacts:
            assert np.hypot(pos[0] - cx, pos[1] - cy) <= r + 1e-6 and abs(pos[2] - exp["z"]) < max(1e-6, tol), (name, backend, pos)


@pytest.mark.parametrize("case", CASES, ids=lambda f: f.__name__)
def test_closed_form_contacts_oracle_and_emulation(model, emul_lib, case):
    q, geoms, exp = case(model)
    for backend, contacts in _backends(model, q, geoms, emul_lib, gpu=False):
        _check(case.__name__, backend, contacts, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda f: f.__name__)
def test_closed_form_contacts_on_gpu(model, case):
    q, geoms, exp = case(model)
    _check(case.__name__, "hip", _contacts_engine(model, q, geoms, None), exp)
