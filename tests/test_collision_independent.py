"""Independent cross-checks of the collision routines that oracle and engine share as algorithm text, on RANDOM
configurations: nothing here reuses their code.

* box-box: a brute-force separating-axis test in numpy (15 axes). For pairs pushed a fraction of a millimetre into each other the deepest
  contact must sit at minus the smallest overlap, and the contact normal must be that axis, pointing from the first geom to the second.
* MPR (cylinder-box, cylinder-cylinder; MuJoCo 2.0 sends cylinders through libccd's MPR [3P] as well): the exact penetration depth of two
  convex bodies is the distance from the origin to the boundary of their Minkowski difference (scipy ConvexHull of the vertex differences,
  cylinders as 96-gon prisms). MPR reads depth and normal off its last portal, so it may report slightly more than the minimum; it must
  never report less, and its normal must be the nearest face's to within a few degrees.

Oracle and lane emulation here; the HIP kernels see the same configurations under -m gpu."""
import numpy as np
import pytest
from scipy.spatial import ConvexHull

from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from oracle.oracle import Oracle
from test_collision_kat import BOX, CYL, MANY, _geoms, _parked


@pytest.fixture(scope="module")
def model():
    return load_model(MANY)


def _rot(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def _sat(pa, Ra, ha, pb, Rb, hb):
    """Overlaps along the 6 face normals and the 9 edge-edge cross products; returns (overlap, axis pointing from a to b, index) of the axis
    a box-box routine with the usual face preference must pick -- an edge-edge axis only wins when its overlap is below 95 % of the best
    face's (oracle / engine: 5 % and 1e-6, like ODE's and MuJoCo's fudge factor [3P]) -- or None when the choice is within 2 % of that line
    or two candidate axes tie."""
    axes = [Ra[:, i] for i in range(3)] + [Rb[:, i] for i in range(3)]
    for i in range(3):
        for j in range(3):
            c = np.cross(Ra[:, i], Rb[:, j])
            axes.append(c / np.linalg.norm(c) if np.linalg.norm(c) > 1e-6 else None)
    ovs = []
    for ax in axes:
        if ax is None:
            ovs.append(np.inf)
            continue
        ra = sum(abs(ax @ Ra[:, i]) * ha[i] for i in range(3))
        rb = sum(abs(ax @ Rb[:, i]) * hb[i] for i in range(3))
        ovs.append(ra + rb - abs(ax @ (pb - pa)))
    ovs = np.array(ovs)
    kf, ke = int(np.argmin(ovs[:6])), 6 + int(np.argmin(ovs[6:]))
    edge_wins = ovs[ke] < 0.95 * ovs[kf] - 1e-6
    k = ke if edge_wins else kf
    ambiguous = abs(ovs[ke] - 0.95 * ovs[kf]) < 0.02 * ovs[kf] + 2e-6            # too close to the preference line
    group = ovs[6:] if edge_wins else ovs[:6]
    ambiguous |= np.sort(group)[1] - np.sort(group)[0] < 2e-5                      # two axes of the winning kind tie
    ax = axes[k] if axes[k] @ (pb - pa) > 0 else -axes[k]
    return float(ovs[k]), ax, k, bool(ambiguous), float(min(ovs.min(), 1e9))


def _box_pts(p, R, h):
    return np.array([p + R @ (np.array([sx, sy, sz]) * h) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])


def _cyl_pts(p, R, r, hh, n=96):
    a = np.linspace(0, 2 * np.pi, n, endpoint=False)
    return np.array([p + R @ np.array([r * np.cos(t), r * np.sin(t), s * hh]) for t in a for s in (-1, 1)])


def _minkowski_depth(A, B):
    """penetration depth (> 0: overlapping) and outward normal of conv(B) - conv(A) at the face nearest to the origin"""
    H = ConvexHull((B[None, :, :] - A[:, None, :]).reshape(-1, 3))
    dist = -H.equations[:, 3]
    k = int(np.argmin(dist))
    return float(dist[k]), H.equations[k, :3]


def _push_together(depth_of, rng, lo=0.0, hi=0.25):
    """distance along a ray at which the pair overlaps by a random 0.2 .. 2 mm (bisection on the independent depth function)"""
    target = rng.uniform(2e-4, 2e-3)
    for _ in range(44):
        mid = 0.5 * (lo + hi)
        if depth_of(mid) > target:
            lo = mid
        else:
            hi = mid
    return lo


class _Backends:
    def __init__(self, model, lib_path, use_engine=True):
        self.m = model
        self.o = Oracle(model)
        self.sim = BatchSim(model, 1, lib_path=lib_path) if use_engine else None

    def contacts(self, q, center):
        self.o.set_state(qpos=q, qvel=np.zeros(self.m.nv))
        self.o.forward()
        near = lambda c: np.abs(np.asarray(c[1:4]) - center).max() < 0.3
        out = {"oracle": [c for c in self.o.contacts() if near(c)]}
        if self.sim is not None:
            self.sim.set_state(qpos=q[None], qvel=np.zeros((1, self.m.nv)))
            d = self.sim.forward_debug()
            assert self.sim.counters()["status"][0] == 0
            out["engine"] = [c for c in d["contacts"][0][:int(d["ncon"][0])] if near(c)]
        return out


def _random_pose(rng, aligned=False):
    if aligned:                                                                 # a few degrees off a face-to-face pose: multi-point face contacts
        ax = rng.normal(size=3)
        ang = rng.uniform(0, 0.06)
        q = np.array([np.cos(ang / 2), *(np.sin(ang / 2) * ax / np.linalg.norm(ax))])
    else:
        q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def _check_box_box(model, be, n_random, n_aligned, seed):
    q0, adr = _parked(model)
    boxes = _geoms(model, BOX)
    rng = np.random.default_rng(seed)
    centre = np.array([0.0, -0.6, 2.0])                                         # mid-air: nothing else within metres
    faces = edges = multi = 0
    for t in range(n_random + n_aligned):
        ga, gb = (int(g) for g in rng.choice(boxes, 2, replace=False))
        qa, qb = _random_pose(rng, t >= n_random), _random_pose(rng, t >= n_random)
        Ra, Rb, ha, hb = _rot(qa), _rot(qb), model.geom_size[ga], model.geom_size[gb]
        d = rng.normal(size=3) if t < n_random else Ra[:, rng.integers(3)] * rng.choice([-1, 1]) + 0.05 * rng.normal(size=3)
        d /= np.linalg.norm(d)
        dist = _push_together(lambda s: _sat(centre, Ra, ha, centre + s * d, Rb, hb)[4], rng)
        pb = centre + dist * d
        ov, ax, k, ambiguous, _ = _sat(centre, Ra, ha, pb, Rb, hb)
        if ambiguous:
            continue                                                            # either answer is right
        q = q0.copy()
        q[adr[ga]:adr[ga] + 7] = [*centre, *qa]
        q[adr[gb]:adr[gb] + 7] = [*pb, *qb]
        sign = 1.0 if ga < gb else -1.0                                          # contacts are reported geom1 < geom2, normal from geom1 to geom2
        # face contact: the deepest corner of the incident box; clipping against the reference face's rim makes the contact shallower than the
        # overlap when that corner projects outside the face
        vertex_inside = True
        if k < 6:
            ref_p, ref_R, ref_h, inc = (centre, Ra, ha, _box_pts(pb, Rb, hb)) if k < 3 else (pb, Rb, hb, _box_pts(centre, Ra, ha))
            v = inc[np.argmin(inc @ ax)] if k < 3 else inc[np.argmax(inc @ ax)]
            local = ref_R.T @ (v - ref_p)
            vertex_inside = all(abs(local[j]) <= ref_h[j] - 1e-6 for j in range(3) if j != k % 3)
        for name, cs in be.contacts(q, centre).items():
            assert len(cs) >= 1, (name, t)
            deepest = min(c[0] for c in cs)
            assert deepest > -ov - 1e-9, (name, t, deepest, ov, k)               # never deeper than the overlap along the chosen axis
            if vertex_inside:                                                    # ... and exactly that deep unless the deepest corner hangs over the rim
                assert abs(deepest + ov) < 1e-9, (name, t, deepest, ov, k)
            for c in cs:
                assert np.abs(np.asarray(c[4:7]) - sign * ax).max() < 1e-7, (name, t, c[4:7], ax)
                assert c[0] <= model.geom_margin[ga] + model.geom_margin[gb] + 1e-12
                for p, R, h in ((centre, Ra, ha), (pb, Rb, hb)):               # every contact point lies in both boxes, up to depth + margin
                    local = R.T @ (np.asarray(c[1:4]) - p)
                    assert (np.abs(local) <= h + ov + 2.5e-3).all(), (name, t, local, h)
        faces += k < 6
        edges += k >= 6
        multi += len(cs) > 1
    return faces, edges, multi


def test_box_box_against_a_brute_force_separating_axis_test(model, emul_lib):
    faces, edges, multi = _check_box_box(model, _Backends(model, emul_lib), 40, 24, 0)
    assert faces >= 20 and edges >= 8 and multi >= 10                           # both SAT branches and the multi-point clipping were exercised


def _check_mpr(model, be, n_box, n_cyl, seed):
    q0, adr = _parked(model)
    boxes, cyls = _geoms(model, BOX), _geoms(model, CYL)
    rng = np.random.default_rng(seed)
    centre = np.array([0.0, -0.6, 2.0])
    worst_cos, worst_ratio = 1.0, 1.0
    for t in range(n_box + n_cyl):
        gc = int(rng.choice(cyls))
        gb = int(rng.choice(boxes)) if t < n_box else int(rng.choice([g for g in cyls if g != gc]))
        qa, qb = _random_pose(rng), _random_pose(rng)
        Ra, Rb = _rot(qa), _rot(qb)
        A = _cyl_pts(centre, Ra, model.geom_size[gc][0], model.geom_size[gc][1])
        other = (lambda p: _box_pts(p, Rb, model.geom_size[gb])) if t < n_box else (lambda p: _cyl_pts(p, Rb, model.geom_size[gb][0], model.geom_size[gb][1]))
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        dist = _push_together(lambda s: _minkowski_depth(A, other(centre + s * d))[0], rng)
        pb = centre + dist * d
        depth, n = _minkowski_depth(A, other(pb))
        q = q0.copy()
        q[adr[gc]:adr[gc] + 7] = [*centre, *qa]
        q[adr[gb]:adr[gb] + 7] = [*pb, *qb]
        for name, cs in be.contacts(q, centre).items():
            assert len(cs) == 1, (name, t, len(cs))                             # MPR: one contact per pair, as in MuJoCo 2.0
            got = -cs[0][0]
            # the 96-gon under-estimates a cylinder's reach by up to r (1 - cos(pi / 96)) = 1.3e-5 m per body
            assert depth - 1e-6 <= got <= 1.06 * depth + 4e-5, (name, t, got, depth)
            # n is the outward normal of conv(B) - conv(A) at the face nearest to the origin: B leaves A along -n. The contact normal points from
            # the first geom of the reported pair to the second (the oracle reports its own geom ids, the engine's dump only the frame)
            first_is_a = int(cs[0][7]) == gc if name == "oracle" else None
            cosn = float(np.asarray(cs[0][4:7]) @ -n)
            if first_is_a is None:
                cosn = abs(cosn)
            elif not first_is_a:
                cosn = -cosn
            assert cosn > 0.995, (name, t, cosn)
            worst_cos, worst_ratio = min(worst_cos, cosn), max(worst_ratio, got / depth)
    return worst_cos, worst_ratio


def test_mpr_depth_against_the_exact_minkowski_penetration(model, emul_lib):
    _check_mpr(model, _Backends(model, emul_lib), 16, 8, 1)


def _qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def test_contacts_are_covariant_under_rigid_motions(model, emul_lib):
    """Frame invariance: moving BOTH bodies of a pair by the same random rigid motion must move the contact set with them -- same depths, points
    and normals carried along. Holds for every pair routine (box-box clipping, MPR for cylinders, the analytic capsule and sphere cases) and
    needs no second implementation: a handedness slip, a world-axis assumption or an un-rotated offset in any of them breaks it."""
    from test_collision_kat import CAP
    q0, adr = _parked(model)
    free = sorted(adr)                                                          # geoms on free joints: spheres, boxes, cylinders, capsules
    rng = np.random.default_rng(5)
    centre = np.array([0.0, -0.6, 2.0])
    be = _Backends(model, emul_lib)
    kinds = set()
    done = 0
    while done < 40:
        ga, gb = (int(g) for g in rng.choice(free, 2, replace=False))
        qa, qb = _random_pose(rng), _random_pose(rng)
        reach = model.geom_rbound[ga] + model.geom_rbound[gb]
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)

        def place(dist):
            q = q0.copy()
            q[adr[ga]:adr[ga] + 7] = [*centre, *qa]
            q[adr[gb]:adr[gb] + 7] = [*(centre + dist * d), *qb]
            return q
        target, lo, hi = rng.uniform(2e-4, 2e-3), 0.0, 1.05 * reach                # the depths contacts have in the simulation
        for _ in range(30):
            mid = 0.5 * (lo + hi)
            cs = be.contacts(place(mid), centre)["oracle"]
            if cs and min(c[0] for c in cs) < -target:
                lo = mid
            else:
                hi = mid
        pb = centre + lo * d
        q = place(lo)
        base = be.contacts(q, centre)
        if len(base["oracle"]) == 0:
            continue
        qt = _random_pose(rng)
        T, shift = _rot(qt), rng.uniform(-0.2, 0.2, size=3)
        move = lambda p: centre + T @ (np.asarray(p) - centre) + shift
        q2 = q0.copy()
        q2[adr[ga]:adr[ga] + 7] = [*move(centre), *_qmul(qt, qa)]
        q2[adr[gb]:adr[gb] + 7] = [*move(pb), *_qmul(qt, qb)]
        moved = be.contacts(q2, centre + shift)
        for name in base:
            a, b = base[name], moved[name]
            assert len(a) == len(b), (name, ga, gb, len(a), len(b))
            for c in a:
                want_p, want_n = move(c[1:4]), T @ np.asarray(c[4:7])
                best = min(b, key=lambda e: np.abs(np.asarray(e[1:4]) - want_p).sum())
                # analytic pairs are covariant to 1e-8 in depth (rounding, the 1e-9 tie rules of the clipping code) and to the tolerance of the capsule-box
                # segment search in point and normal; MPR (any pair with a cylinder) stops at a 1e-6 portal tolerance, which it may reach
                # along another sequence of portals in the moved frame: depth to ~1e-7, point to ~1e-5, normal to ~5e-3 (test_many_objects.py)
                td, tp, tn = (5e-6, 1e-4, 2e-2) if CYL in (int(model.geom_type[ga]), int(model.geom_type[gb])) else (1e-8, 1e-7, 1e-6)
                assert abs(best[0] - c[0]) < td and np.abs(np.asarray(best[1:4]) - want_p).max() < tp and np.abs(np.asarray(best[4:7]) - want_n).max() < tn, (
                    name, ga, gb, int(model.geom_type[ga]), int(model.geom_type[gb]), c[:7], best[:7])
        kinds.add((int(model.geom_type[ga]), int(model.geom_type[gb])))
        done += 1
    assert len(kinds) >= 8, kinds                                               # most type combinations occurred


def test_contact_dynamics_are_covariant_under_motions_that_keep_gravity(model, emul_lib):
    """One forward pass of two free objects colliding in mid-air, with random velocities, before and after a half turn about the vertical plus
    a shift: linear accelerations (world frame) must turn along, angular ones (body frame) must stay -- through contact Jacobians, friction
    pyramids, the coupled Newton Hessian and its solve -- and the contact's forces on the two bodies must cancel. No second implementation involved. (Only the half turn is an exact symmetry: the
    friction pyramid's tangents are built from the world y / z axis as in MuJoCo's mju_makeFrame [3P], so other angles orient the 4-sided
    pyramid differently; under a half turn the tangents just change sign, and the pyramid has both signs.)"""
    q0, adr = _parked(model)
    free = sorted(adr)
    rng = np.random.default_rng(7)
    centre = np.array([0.0, -0.6, 2.0])
    be = _Backends(model, emul_lib)
    vadr = {g: int(model.jnt_dofadr[model.body_jntadr[int(model.geom_bodyid[g])]]) for g in free}
    done = acting = 0
    while done < 16:
        ga, gb = (int(g) for g in rng.choice(free, 2, replace=False))
        if CYL in (int(model.geom_type[ga]), int(model.geom_type[gb])):
            continue                                                            # MPR normals are only reproducible to ~5e-3 between frames (above)
        qa, qb = _random_pose(rng), _random_pose(rng)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        reach = model.geom_rbound[ga] + model.geom_rbound[gb]
        target, lo, hi = rng.uniform(3e-4, 1.5e-3), 0.0, 1.05 * reach

        def state(dist, ang=0.0, shift=np.zeros(3), vel=None):
            qz = np.array([np.cos(ang / 2), 0, 0, np.sin(ang / 2)]) if ang != np.pi else np.array([0.0, 0.0, 0.0, 1.0])
            Rz = _rot(qz)
            q, v = q0.copy(), np.zeros(model.nv)
            for g, p, quat in ((ga, centre, qa), (gb, centre + dist * d, qb)):
                q[adr[g]:adr[g] + 7] = [*(centre + Rz @ (p - centre) + shift), *_qmul(qz, quat)]
            if vel is not None:
                for g, w in ((ga, vel[:6]), (gb, vel[6:])):
                    v[vadr[g]:vadr[g] + 3] = Rz @ w[:3]
                    v[vadr[g] + 3:vadr[g] + 6] = w[3:]
            return q, v, Rz
        for _ in range(30):
            mid = 0.5 * (lo + hi)
            cs = be.contacts(state(mid)[0], centre)["oracle"]
            if cs and min(c[0] for c in cs) < -target:
                lo = mid
            else:
                hi = mid
        if not be.contacts(state(lo)[0], centre)["oracle"]:
            continue
        vel = np.concatenate([rng.normal(size=3) * 0.2, rng.normal(size=3) * 2.0, rng.normal(size=3) * 0.2, rng.normal(size=3) * 2.0])
        ang, shift = np.pi, np.array([*rng.uniform(-0.2, 0.2, size=2), 0.0])
        acc = {}
        for tag, (q, v, Rz) in (("a", state(lo, 0.0, np.zeros(3), vel)), ("b", state(lo, ang, shift, vel))):
            be.o.set_state(qpos=q, qvel=v)
            be.o.forward()
            acc["oracle", tag] = be.o.vec("qacc").copy()
            if tag == "a":                                                      # Newton's third law: the contact's net force on the pair vanishes
                da = acc["oracle", tag] - be.o.vec("qacc_smooth")
                net = sum((model.body_mass[int(model.geom_bodyid[g])] + model.dof_armature[vadr[g]]) * da[vadr[g]:vadr[g] + 3] for g in (ga, gb))
                each = max(np.abs(model.body_mass[int(model.geom_bodyid[g])] * da[vadr[g]:vadr[g] + 3]).max() for g in (ga, gb))
                assert np.abs(net).max() < 1e-9 * max(1.0, each), (ga, gb, net, each)
                acting += each > 0.1                                               # (a pair whose random velocities separate it feels nothing)
            be.sim.set_state(qpos=q[None], qvel=v[None], warmstart=np.zeros((1, model.nv)))
            acc["engine", tag] = be.sim.forward_debug()["qacc"][0][:model.nv].copy()
        Rz = state(lo, ang)[2]
        for name in ("oracle", "engine"):
            a, b = acc[name, "a"], acc[name, "b"]
            scale = max(1.0, np.abs(a).max())
            for g in (ga, gb):
                i = vadr[g]
                assert np.abs(Rz @ a[i:i + 3] - b[i:i + 3]).max() < 1e-7 * scale, (name, ga, gb, a[i:i + 3], b[i:i + 3])
                assert np.abs(a[i + 3:i + 6] - b[i + 3:i + 6]).max() < 1e-7 * scale, (name, ga, gb, a[i + 3:i + 6], b[i + 3:i + 6])
        done += 1
    assert acting >= 8, acting


@pytest.mark.gpu
def test_independent_collision_checks_on_gpu(model):
    be = _Backends(model, None)
    faces, edges, multi = _check_box_box(model, be, 14, 6, 2)
    assert faces >= 4 and edges >= 2
    _check_mpr(model, be, 3, 2, 3)
