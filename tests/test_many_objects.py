"""The many-object engine variant (csrc/ur5sim_many.hip, UR5gripper_2_finger_many_objects.xml: 40 free objects, condim 6,
nv = 248) against the oracle: lane-emulation build on CPU, the real kernel with -m gpu.

Parity horizon: piles of cylinders / capsules resting on single MPR contacts are chaotic, and MPR itself is discontinuous at
its 1e-6 tolerance (a 1e-14 state difference can flip a portal step and move a normal by 5e-3), so trajectories are compared
over the first contacts of the drop (tens of steps, 1e-9) and statistically afterwards.
"""
import os
import sys

import numpy as np
import pytest

from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from oracle.oracle import Oracle, lib as olib

MANY = "/UR5+gripper/UR5gripper_2_finger_many_objects.xml"


@pytest.fixture(scope="module")
def model_many():
    return load_model(MANY)


def _drop_parity(model, sim, scene, seed, nsteps, tol):
    """Step-by-step parity along the oracle's trajectory: before every step the engine is put into the oracle's state, both step
    once, the results are compared. Without the re-synchronisation a single MPR branch flip (rounding-level input differences decide
    which portal face is refined) would end the comparison for good; with it a flip costs one step, and at most one is tolerated."""
    o = Oracle(model)
    o.reset(seed, 1, False)
    assert np.array_equal(o.get_state()["qpos"], sim.get_state()["qpos"][scene])   # same reset distribution (GraspingEnv.py:420-430)
    errs, contacts_seen = [], 0
    for k in range(nsteps):
        st = o.get_state()
        sim.set_state(qpos=st["qpos"][None], qvel=st["qvel"][None], warmstart=st["warmstart"][None], pid=st["pid"][None])
        o.step(1)
        sim.step(1)
        contacts_seen = max(contacts_seen, olib().ur5o_ncon(o._h))
        a, b = o.get_state(), sim.get_state()
        errs.append(max(np.abs(a["qpos"] - b["qpos"][scene]).max(), 1e-2 * np.abs(a["qvel"] - b["qvel"][scene]).max()))
    errs = np.array(errs)
    assert contacts_seen >= 2                                        # the comparison window does contain contacts
    assert (errs > tol).sum() <= 1 and errs.max() < 1e-3, errs


def test_drop_matches_oracle_emul(model_many, emul_lib):
    sim = BatchSim(model_many, 1, lib_path=emul_lib)
    sim.reset([20], 1, 0.0)
    _drop_parity(model_many, sim, 0, 20, 60, 1e-9)


def test_forward_quantities_match_oracle_in_a_settled_pile(model_many, emul_lib):
    """After 0.6 s the pile is dense (30+ contacts, coupled bodies): one forward pass of both from the SAME state must agree on
    contacts and on the constrained acceleration (envelope Cholesky vs the oracle's dense Newton)."""
    sim = BatchSim(model_many, 1, lib_path=emul_lib)
    sim.reset([21], 1, 0.0)
    sim.step(300)
    st = sim.get_state()
    o = Oracle(model_many)
    o.set_state(qpos=st["qpos"][0], qvel=st["qvel"][0], warmstart=st["warmstart"][0], pid=st["pid"][0])
    o.forward()
    d = sim.forward_debug()
    oc = o.contacts()
    assert d["ncon"][0] == len(oc) and len(oc) >= 20
    ec = d["contacts"][0][:len(oc)]
    for c in oc:                                                      # contact order differs (pair order vs slot claiming)
        best = min(ec, key=lambda e: np.abs(e[1:4] - c[1:4]).sum())
        assert np.abs(best[1:4] - c[1:4]).max() < 1e-9 and np.abs(best[4:7] - c[4:7]).max() < 1e-9 and abs(best[0] - c[0]) < 1e-9
    qacc = o.vec("qacc")
    assert np.abs(d["qacc"][0][:model_many.nv] - qacc).max() < 1e-6 * max(1.0, np.abs(qacc).max())
    assert sim.counters()["status"][0] == 0


def test_envelope_structure_equals_its_numpy_restatement(model_many, emul_lib):
    """csrc/ur5_engine.h envelope_structure() -- islands of the coupling graph, block order (island, then x, robot last), first coupled block of every block,
    envelope size -- against tools/pile_structure_stats.py, a numpy restatement that only sees the oracle's contacts and body positions: coupled contacts,
    coupled blocks and envelope doubles equal on piles at two stages of the drop. (The structure only decides where the Newton Hessian is stored and in
    which order it is factored; that the factor is right is the qacc parity of the tests above.)"""
    import ctypes as C, os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import pile_structure_stats as P
    from mujoco_rl_ur5_amd.native import _VARIANT
    m = model_many
    blk, nobj, obj_body = P.blocks_of_bodies(m)
    assert nobj == 40 and (blk == nobj).sum() >= 8
    sim = BatchSim(m, 1, lib_path=emul_lib)
    stride = _VARIANT[sim.variant][1]
    seen = 0
    for seed, ms in ((20, 300.0), (23, 500.0)):
        o = Oracle(m)
        o.reset(seed, 1, False)
        o.stay(ms)
        o.forward()
        st = o.get_state()
        pairs = []
        for c in o.contacts():
            a, b = blk[int(m.geom_bodyid[int(c[7])])], blk[int(m.geom_bodyid[int(c[8])])]
            if a >= 0 and b >= 0 and a != b:
                pairs.append((a, b))
        s = P.structure(nobj, o.body_xpos()[obj_body, 0], pairs)
        sim.set_state(qpos=st["qpos"][None], qvel=st["qvel"][None], warmstart=st["warmstart"][None], pid=st["pid"][None])
        out = np.zeros((1, stride))
        assert sim.lib.ur5_forward_debug(sim._h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert (len(pairs), s["coupled"], s["env"]) == (int(out[0, 5]), int(out[0, 6]), int(out[0, 4])), (seed, ms, len(pairs), s["coupled"], s["env"], out[0, 4:7])
        seen += len(pairs)
    assert seen >= 8


def test_envelope_in_global_memory_path(model_many):
    """Envelopes larger than UR5_HENV_CAP doubles run the same factorisation on the scene's global-memory scratch; a build
    with a tiny cap forces that path for every step."""
    from conftest import build_emul
    sim = BatchSim(model_many, 1, lib_path=build_emul(("-DUR5_HENV_CAP=8",), "libur5sim_emul_globalenv.so"))
    sim.reset([20], 1, 0.0)
    _drop_parity(model_many, sim, 0, 20, 30, 1e-9)
    sim.step(270)
    st = sim.get_state()
    o = Oracle(model_many)
    o.set_state(qpos=st["qpos"][0], qvel=st["qvel"][0], warmstart=st["warmstart"][0], pid=st["pid"][0])
    o.forward()
    d = sim.forward_debug()
    qacc = o.vec("qacc")
    assert d["ncon"][0] >= 20 and np.abs(d["qacc"][0][:model_many.nv] - qacc).max() < 1e-6 * max(1.0, np.abs(qacc).max())
    assert d["env_in_lds"][0] == 0 and d["dcache_in_lds"][0] == 0 and d["envelope_doubles"][0] > 8


def test_envelope_and_block_cache_live_in_the_lds_pool(model_many, emul_lib):
    """Round 5: the envelope of the Newton Hessian (and the cache of its factored diagonal blocks) sits in LDS arrays that are dead during assembly / factorisation /
    solve -- kinematic temporaries | aref offsets | search-direction images, 2 384 doubles -- whenever it fits; it does for the whole drop and for the settled pile.
    Same arithmetic as the global-scratch path: the two builds agree bit for bit, and both agree with the oracle."""
    from conftest import build_emul
    sims = [BatchSim(model_many, 1, lib_path=emul_lib), BatchSim(model_many, 1, lib_path=build_emul(("-DUR5_HENV_CAP=8",), "libur5sim_emul_globalenv.so"))]
    modes = []
    for sim in sims:
        sim.reset([21], 1, 0.0)
        m = []
        for _ in range(6):
            sim.step(100)
            d = sim.forward_debug()
            m.append((int(d["env_in_lds"][0]), int(d["dcache_in_lds"][0]), int(d["envelope_doubles"][0]), int(d["ncon"][0])))
        modes.append(m)
    a, b = sims[0].get_state(), sims[1].get_state()
    assert all(np.array_equal(a[k], b[k]) for k in ("qpos", "qvel", "warmstart"))            # 600 steps, contacts from step ~200 on: identical bits
    assert all(e == 1 and dc == 1 and 861 <= tot <= 1856 for e, dc, tot, _ in modes[0]) and all(e == 0 and dc == 0 for e, dc, _, _ in modes[1])
    assert [x[2:] for x in modes[0]] == [x[2:] for x in modes[1]] and modes[0][-1][3] >= 20


def test_render_with_cylinders_and_capsules(model_many, emul_lib):
    sim = BatchSim(model_many, 1, lib_path=emul_lib)
    sim.reset([20], 1, 0.0)
    sim.step(300)
    st = sim.get_state()
    o = Oracle(model_many)
    o.set_state(qpos=st["qpos"][0], qvel=st["qvel"][0])
    cam = model_many.camera_name2id("top_down")
    rgb, depth = sim.render(cam, 120, 120, 0)
    rgbo, deptho = o.render(cam, 120, 120, 0)
    assert (np.abs(depth[0] - deptho) > 1e-4).mean() < 2e-3           # fp32 vs fp64: a few grazing-ray pixels on edge-on plates
    assert (np.abs(rgb[0].astype(int) - rgbo.astype(int)).max(axis=2) > 2).mean() < 4e-3


@pytest.mark.gpu
def test_many_object_kernel_matches_oracle(model_many):
    import torch
    assert torch.cuda.is_available()
    sim = BatchSim(model_many, 4)
    assert sim.variant == 1
    sim.reset(20 + np.arange(4, dtype=np.uint64), 1, 0.0)
    _drop_parity(model_many, sim, 2, 22, 60, 1e-9)


@pytest.mark.gpu
def test_many_object_kernel_settles_piles(model_many):
    sim = BatchSim(model_many, 64)
    sim.reset(100 + np.arange(64, dtype=np.uint64), 1, 1000.0)         # reset_model: drop + stay(1000), GraspingEnv.py:473
    c = sim.counters()
    st = sim.get_state()
    z = st["qpos"][:, 8:].reshape(64, -1, 7)[:, :, 2]
    assert (c["status"] == 0).all() and (c["total_steps"] == 491).all()
    assert np.isfinite(st["qpos"]).all() and (z > -0.01).all() and (z < 1.25).all()   # objects that miss the bins lie on the ground
    assert (c["ncon_max"] > 20).all() and (c["ncon_max"] < 160).all()
    r, steps = sim.move_group(1 << 6, [[0.2]], 0.05, 300)             # the gripper still obeys its PID with the pile present
    assert (r == 0).all()


def _isolate(model, which_type, quat, z_above_floor):
    """State in which one object of geom type `which_type` sits alone on the pick-bin floor and every other object is parked far away
    on the ground plane (spread out so that nothing else touches anything of interest). Returns (qpos, geom id, qpos address)."""
    o = Oracle(model)
    o.reset(20, 1, False)
    q = o.get_state()["qpos"].copy()
    gt, gb = np.asarray(model.geom_type), np.asarray(model.geom_bodyid)
    g = int(np.where(gt == which_type)[0][0])
    floor_top = 0.89                                                     # UR5gripper_2_finger_many_objects.xml:120 (pick_box plate)
    adr_of = {}
    for gi in range(model.ngeom):
        b = int(gb[gi])
        if model.body_jntnum[b] == 1 and model.jnt_type[model.body_jntadr[b]] == 0:
            adr_of[gi] = int(model.jnt_qposadr[model.body_jntadr[b]])
    for k, (gi, adr) in enumerate(sorted(adr_of.items())):
        q[adr:adr + 7] = [3.0 + 0.4 * (k % 8), 2.0 + 0.4 * (k // 8), 0.2, 1, 0, 0, 0]
    adr = adr_of[g]
    r = float(model.geom_size[g][0])
    q[adr:adr + 7] = [0.0, -0.6, floor_top + r + z_above_floor, *quat]
    return q, g, adr


def test_capsule_lying_on_the_bin_floor_gets_two_contacts(model_many):
    """mjc_CapsuleBox [3P] gives a capsule lying on a face one contact under each end sphere; so do oracle and engine (own restatement),
    which keeps it from rocking on a single MPR point. Normal forces: m g / 2 each once it has settled."""
    s = np.sqrt(0.5)
    q, g, adr = _isolate(model_many, 3, [s, 0, s, 0], -2e-4)              # axis along world x, 0.2 mm into the plate
    o = Oracle(model_many)
    o.set_state(qpos=q, qvel=np.zeros(model_many.nv))
    o.forward()
    mine = [c for c in o.contacts() if int(c[7]) == g or int(c[8]) == g]
    assert len(mine) == 2
    h = float(model_many.geom_size[g][1])
    xs = sorted(c[1] for c in mine)
    assert abs(xs[0] + h) < 1e-6 and abs(xs[1] - h) < 1e-6             # under the two end spheres
    for c in mine:
        assert abs(abs(c[6]) - 1) < 1e-9 and abs(c[0] + 2e-4) < 1e-9     # vertical normal, the prescribed penetration
    o.step(400)
    st = o.get_state()
    assert np.abs(st["qvel"][adr - 8 - (adr - 8) // 7: adr - 8 - (adr - 8) // 7 + 6]).max() < 1e-3   # at rest (dof address = qpos address - #objects before it)
    o.forward()
    mine = [c for c in o.contacts() if int(c[7]) == g or int(c[8]) == g]
    b = int(model_many.geom_bodyid[g])
    mg = float(model_many.body_mass[b]) * 9.81
    assert len(mine) == 2 and all(abs(c[10] - 0.5 * mg) < 0.02 * mg for c in mine)


def test_capsule_primitives_engine_equals_oracle(model_many, emul_lib):
    """A pose that exercises sphere-capsule, capsule-capsule (crossed and parallel) and capsule-box together: same contacts from both."""
    o = Oracle(model_many)
    o.reset(20, 1, False)
    q = o.get_state()["qpos"].copy()
    gt, gb = np.asarray(model_many.geom_type), np.asarray(model_many.geom_bodyid)
    adr = lambda gi: int(model_many.jnt_qposadr[model_many.body_jntadr[int(gb[gi])]])
    caps = [int(x) for x in np.where(gt == 3)[0][:3]]
    sph = int(np.where(gt == 2)[0][0])
    for k, gi in enumerate(gi for gi in range(model_many.ngeom) if model_many.body_jntnum[int(gb[gi])] == 1 and model_many.jnt_type[model_many.body_jntadr[int(gb[gi])]] == 0):
        q[adr(gi):adr(gi) + 7] = [3.0 + 0.4 * (k % 8), 2.0 + 0.4 * (k // 8), 0.2, 1, 0, 0, 0]
    s = np.sqrt(0.5)
    r = [float(model_many.geom_size[c][0]) for c in caps]
    q[adr(caps[0]):adr(caps[0]) + 7] = [0.0, -0.6, 0.89 + r[0] - 1e-4, s, 0, s, 0]                      # lying along x on the floor
    q[adr(caps[1]):adr(caps[1]) + 7] = [0.0, -0.6, 0.89 + 2 * r[0] + r[1] - 3e-4, s, s, 0, 0]           # across it, along y
    q[adr(caps[2]):adr(caps[2]) + 7] = [0.01, -0.6 + r[0] + r[2] - 2e-4, 0.89 + r[0] - 1e-4, s, 0, s, 0]  # beside it at axis height, parallel, shifted 1 cm
    q[adr(sph):adr(sph) + 7] = [0.0, -0.6 - r[0] - float(model_many.geom_size[sph][0]) + 2e-4, 0.89 + r[0], 1, 0, 0, 0]   # sphere touching its side
    o.set_state(qpos=q, qvel=np.zeros(model_many.nv))
    o.forward()
    sim = BatchSim(model_many, 1, lib_path=emul_lib)
    sim.set_state(qpos=q[None], qvel=np.zeros((1, model_many.nv)))
    d = sim.forward_debug()
    oc = o.contacts()
    touched = {(int(c[7]), int(c[8])) for c in oc}
    assert {tuple(sorted((caps[0], caps[1]))), tuple(sorted((caps[0], caps[2]))), (sph, caps[0])} <= {tuple(sorted(t)) if t != (sph, caps[0]) else t for t in touched}
    assert d["ncon"][0] == len(oc)
    ec = d["contacts"][0][:len(oc)]
    for c in oc:
        best = min(ec, key=lambda e: np.abs(e[1:4] - c[1:4]).sum())
        assert np.abs(best[1:4] - c[1:4]).max() < 1e-12 and np.abs(best[4:7] - c[4:7]).max() < 1e-12 and abs(best[0] - c[0]) < 1e-12


def _forward_agrees(model, sim, min_robot_contacts):
    st = sim.get_state()
    o = Oracle(model)
    o.set_state(qpos=st["qpos"][0], qvel=st["qvel"][0], warmstart=st["warmstart"][0], pid=st["pid"][0])
    o.set_ctrl(sim.get_ctrl()[0])
    o.forward()
    d = sim.forward_debug()
    oc = o.contacts()
    gb = np.asarray(model.geom_bodyid)
    robot_bodies = {b for b in range(model.nbody) if model.body_treeid[b] == 0}
    nrobot = sum(1 for c in oc if int(gb[int(c[7])]) in robot_bodies or int(gb[int(c[8])]) in robot_bodies)
    assert nrobot >= min_robot_contacts, nrobot
    assert d["ncon"][0] == len(oc)
    qacc = o.vec("qacc")
    assert np.abs(d["qacc"][0][:model.nv] - qacc).max() < 1e-6 * max(1.0, np.abs(qacc).max())
    return nrobot


def test_gripper_in_the_pile_couples_robot_and_objects(model_many, emul_lib):
    """The robot block (8 wide, last in the envelope) coupled with object blocks: push the open gripper into the settled pile and
    compare one forward pass with the oracle's dense Newton from the same state."""
    sim = BatchSim(model_many, 1, lib_path=emul_lib)
    sim.reset([23], 1, 0.0)
    sim.step(250)
    xy = sim.body_xpos()[0, 8:]
    k = int(np.argmin(np.hypot(xy[:, 0], xy[:, 1] + 0.6)))                # the object nearest to the bin centre
    sim.move_ee([[xy[k, 0], xy[k, 1], 1.1]], 0.05, 400)
    sim.move_ee([[xy[k, 0], xy[k, 1], xy[k, 2] + 0.01]], 0.01, 250)       # fingers around / onto it
    _forward_agrees(model_many, sim, 1)
    sim.move_group(1 << 6, [[-0.4]], 0.01, 150)                           # close on it
    _forward_agrees(model_many, sim, 1)
    assert sim.counters()["status"][0] == 0


def test_closed_gripper_robot_robot_contact(model_many, emul_lib):
    """Finger against finger: a contact between two robot bodies adds its coupling inside the robot block."""
    sim = BatchSim(model_many, 1, lib_path=emul_lib)
    sim.reset([20], 1, 0.0)
    st = sim.get_state()
    q = st["qpos"].copy()
    q[0, 8:] = np.tile([3.0, 3.0, 5.0, 1, 0, 0, 0], 40) + np.repeat(np.arange(40) * 0.3, 7) * np.tile([1, 0, 0, 0, 0, 0, 0], 40)   # objects out of the way
    sim.set_state(qpos=q)
    r, steps = sim.move_group(1 << 6, [[-0.9]], 0.001, 250)               # close far beyond an object-sized gap
    _forward_agrees(model_many, sim, 1)


def _forward_parity_from_engine_state(model, sim, scene, min_contacts):
    """One forward pass of oracle and engine from the SAME (engine) state: same contact set, same constrained acceleration."""
    st = sim.get_state()
    o = Oracle(model)
    o.set_state(qpos=st["qpos"][scene], qvel=st["qvel"][scene], warmstart=st["warmstart"][scene], pid=st["pid"][scene])
    o.set_ctrl(sim.get_ctrl()[scene])
    o.forward()
    d = sim.forward_debug()
    oc = o.contacts()
    assert d["ncon"][scene] == len(oc) and len(oc) >= min_contacts, (d["ncon"][scene], len(oc))
    ec = d["contacts"][scene][:len(oc)]
    # Round 6: BIT-equal. Both lists are in geom-pair order; every contact between objects, and between an object and the bins, has the oracle's distance, position and
    # normal word for word (the pile unit's kinematics and collision follow the oracle's text without fused multiply-adds: tools/contact_bits.py, 3 610 of 3 610 contacts on
    # the MI355X). Rounds 3-5 accepted 1e-8 m / 1e-6 / 1e-10 m here and one portal flip per scene (a 5e-3 jump of a normal, 1 contact in 935: the two texts then differed in
    # the last bit of a quaternion normalisation and of the box-box vertices). Contacts of a ROBOT geom keep those bounds: the arm's kinematics is another text by design.
    gb = np.asarray(model.geom_bodyid)
    flips = 0
    for e, c in zip(ec, oc):
        robot = any(model.body_treeid[int(gb[int(g)])] == 0 for g in (c[7], c[8]))   # the robot's kinematic tree (the bins and the floor have none)
        if robot:
            assert np.abs(e[1:4] - c[1:4]).max() < 1e-8 and np.abs(e[4:7] - c[4:7]).max() < 1e-6 and abs(e[0] - c[0]) < 1e-10, (e, c)
        elif not np.array_equal(e[:7].view(np.uint64), c[:7].view(np.uint64)):
            flips += 1
    assert flips == 0, flips
    qacc = o.vec("qacc")
    err = np.abs(d["qacc"][scene][:model.nv] - qacc).max() / max(1.0, np.abs(qacc).max())
    assert err < 1e-9, err                                            # (round 5: 1e-6; measured 1e-14 on the lane emulation, tools/pile_early_divergence.py has the trajectories)
    return flips


@pytest.mark.gpu
def test_settled_pile_forward_parity_on_gpu(model_many):
    """The HIP many-object kernel in a dense, settled pile (reset_model: drop + 1000 ms): contacts and constrained acceleration of several
    scenes against the oracle started from the kernel's own state (the CPU-emulation twin is test_forward_quantities_match_oracle_...)."""
    sim = BatchSim(model_many, 8)
    sim.reset(300 + np.arange(8, dtype=np.uint64), 1, 1000.0)
    assert sim.counters()["status"].max() == 0
    flips = sum(_forward_parity_from_engine_state(model_many, sim, scene, 30) for scene in range(8))
    assert flips == 0, flips                                              # eight piles, ~300 contacts, every one the oracle's bits


@pytest.mark.gpu
def test_pile_grasp_bits_against_the_oracle_on_gpu(model_many):
    """Grasp attempts on settled piles: the HIP kernel and the oracle start from the SAME settled state (the kernel's), run one full
    move_and_grasp script each, and must agree on the reward bit and the script's result codes. The 24 scenes are the ones of a 256-pile pool
    whose best box (tools/pile_aim.py: level top face, sides parallel to the fingers) scores lowest, so that the statistic HAS positives -- with
    the round-2 rule (any object, rotation e % 6) 2 % of the attempts succeeded and all-zeros on both sides passed. tools/gpu_many_agreement.py
    runs the 256-of-3072-scene statistic kept under profiles/ (28 % positives, 95-97 % agreement, 93-96 % of the oracle's positives reproduced)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from pile_aim import pick_box
    pool, n = 256, 24
    sim = BatchSim(model_many, pool)
    sim.reset(500 + np.arange(pool, dtype=np.uint64), 1, 1000.0)
    st = sim.get_state()
    ctrl = sim.get_ctrl()
    picks = [pick_box(model_many, st["qpos"][e]) for e in range(pool)]
    sel = sorted([e for e in range(pool) if picks[e] is not None], key=lambda e: picks[e][3])[:n]
    assert len(sel) == n and picks[sel[-1]][3] < 25
    acts, rots = np.zeros((pool, 3)), np.zeros(pool, dtype=np.int64)
    acts[:] = [0.0, -0.6, 1.0]
    for e in sel:
        acts[e], rots[e] = picks[e][1], picks[e][2]
    rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0)
    assert sim.counters()["status"].max() == 0

    CK = [10, 20, 40, 60, 80, 100, 120, 160, 240, 400, 640, 1000]            # physics steps into the attempt at which trajectories are compared (below)
    NT = 12                                                                   # scenes that also get a rounding-level twin of the oracle

    def one(job):
        e, twin = job
        o = Oracle(model_many)
        o.set_state(qpos=st["qpos"][e], qvel=st["qvel"][e], warmstart=st["warmstart"][e], pid=st["pid"][e])
        o.set_ctrl(ctrl[e])
        if twin:
            o.set_contact_order(1)                                            # the same contacts in reversed order: only the association order of the sums changes
        o.set_checkpoints(CK)
        r, pso, pro = o.grasp_attempt(acts[e], int(rots[e]), 0)
        ck = np.full((len(CK), model_many.nq), np.nan)
        got = o.get_checkpoints()
        ck[:len(got)] = got
        return r, pro, ck
    with ThreadPoolExecutor(max_workers=min(n + NT, os.cpu_count() or 8)) as ex:
        allres = list(ex.map(one, [(e, False) for e in sel] + [(e, True) for e in sel[:NT]]))
    res = [(r, pro) for r, pro, _ in allres[:n]]
    orew = [r for r, _ in res]
    bits = sum(int(r == rew[e]) for e, (r, _) in zip(sel, res))
    codes = sum(int(pro.tolist() == pr[e].tolist()) for e, (_, pro) in zip(sel, res))
    # WHEN do they part (round-4 verdict 3c)? The end bit cannot see a kernel bug that costs 1 % agreement; the trajectory can. The first NT scenes replayed on the kernel
    # with step caps (ur5_set_step_cap_dev: one frozen copy per checkpoint) against the oracle's samples at the same steps, next to the oracle against its own twin:
    # index of the first checkpoint with max |dqpos| > 1e-6. A kernel of the oracle's rounding family parts from it about when the twin does -- not checkpoints earlier.
    import torch
    K = len(CK)
    cap_sim = BatchSim(model_many, NT * K)
    rep = lambda a: np.repeat(a[sel[:NT]], K, axis=0)
    cap_sim.set_state(qpos=rep(st["qpos"]), qvel=rep(st["qvel"]), warmstart=rep(st["warmstart"]), pid=rep(st["pid"]))
    cap_sim.set_ctrl(rep(ctrl))
    caps = torch.tensor(np.tile(CK, NT), dtype=torch.int32, device="cuda")
    cap_sim.set_step_cap_dev(caps.data_ptr())
    cap_sim.grasp_attempt(rep(acts), rot=rep(rots), check_mode=0)
    gq = cap_sim.get_state()["qpos"].reshape(NT, K, -1)

    def first(a, b):
        d = np.abs(a - b).max(axis=1)                                         # NaN where the oracle's attempt ended before that checkpoint
        bad = np.flatnonzero(d > 1e-6)
        return int(bad[0]) if len(bad) else K
    kernel_idx = [first(gq[i], allres[i][2]) for i in range(NT)]
    twin_idx = [first(allres[n + i][2], allres[i][2]) for i in range(NT)]
    # Round 6: contacts bit-equal to the oracle's from the same state and the position update in the oracle's arithmetic (csrc/ur5_engine_integrate.inc): ten steps in, the
    # kernel is where the oracle's own rounding twins are (1e-15; round 5: 1e-13, bound 1e-7), and it parts from the oracle WHEN the twin does -- the round-5 form of this
    # line granted the kernel one checkpoint of a four-point grid, i.e. a factor 3 in steps. (256 scenes: tools/pile_divergence_time.py, profiles/r06_pile_divergence_time.json.)
    assert np.abs(gq[:, 0] - np.stack([allres[i][2][0] for i in range(NT)])).max() < 1e-12, "ten steps in, kernel and oracle agree to rounding"
    kernel_steps, twin_steps = [(CK + [2 * CK[-1]])[i] for i in kernel_idx], [(CK + [2 * CK[-1]])[i] for i in twin_idx]
    # Twelve scenes are a small sample of a wide distribution (quartiles 60 - 120 over 256 scenes): the suite asks for the verdict's absolute bar -- a median of at least 72
    # steps; round 5: 40 -- and for the twin's median within one step of THIS grid (80 -> 100); the 256-scene statistic with the 0.9 x control-twin criterion is
    # profiles/r06_pile_divergence_time.json (kernel 80 steps, control twins 80).
    assert np.median(kernel_steps) >= 72 and np.median(kernel_steps) >= 0.75 * np.median(twin_steps) and min(kernel_idx) >= 1, (kernel_steps, twin_steps)
    assert sum(orew) >= 2, orew                                               # a statistic with positives (28 % in the 256-scene run)
    # Piles are chaotic, and round 4 measured how chaotic (tools/pile_chaos_floor.py, profiles/r04_pile_chaos_floor_256of3072.json): the ORACLE agrees with its own
    # rounding-level twins -- the same contacts in reversed order, one coordinate moved by 1 ulp -- on 94.5-96.1 % of the grasp bits and 89-91 % of the result codes of
    # 256 such attempts; the kernel (deterministic since round 4: a failing scene can be replayed) agrees with the oracle and its twins on 94.1-95.7 % / 89-91 %. The bound
    # is therefore statistical: at 94.5 % per scene, 5 or more of 24 bits differ with P ~ 1 %; at 90 %, 8 or more of 24 code vectors with P < 1 %.
    assert bits >= n - 4 and codes >= n - 7, (bits, codes, rew[sel].tolist(), orew)


@pytest.mark.gpu
def test_pile_kernel_is_run_to_run_deterministic_on_gpu(model_many):
    """The reference is one thread of MuJoCo: same state, same result (MujocoController.py:379). ur5m_run_kernel spreads a scene over four wavefronts;
    since round 4 its contacts are sorted into geom-pair order and every sum over contacts (body wrenches, twist-space Hessians, coupling blocks) runs
    along fixed lists instead of LDS float atomics, so nothing depends on how the wavefronts were scheduled: the settle of 256 piles (500 steps of falling,
    colliding objects) and a whole grasp round, each run twice from the same records, must agree in EVERY word -- rewards, the 12 phase step counts and result
    codes, and the full state records including the solver-iteration and step counters. (CPU twin: tests/test_engine_simt.py permutes the lane schedule.)"""
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from pile_aim import pick_box
    n = 256
    sim, twin = BatchSim(model_many, n), BatchSim(model_many, n)               # (two handles: a second reset of ONE handle starts from another PID history)
    seeds = 500 + np.arange(n, dtype=np.uint64)
    sim.reset(seeds, 1, 1000.0)
    twin.reset(seeds, 1, 1000.0)
    rec_a = sim.state_tensor("cuda").clone()
    torch.cuda.synchronize()
    assert torch.equal(rec_a, twin.state_tensor("cuda")), "two settles of the same seeds differ"
    twin.close()
    st = sim.get_state()
    acts, rots = np.zeros((n, 3)), np.arange(n) % 6
    acts[:] = [0.0, -0.6, 1.0]
    for e in range(n):
        b = pick_box(model_many, st["qpos"][e])
        if b is not None:
            acts[e], rots[e] = b[1], b[2]
    runs = []
    for _ in range(2):
        sim.state_tensor("cuda").copy_(rec_a)
        torch.cuda.synchronize()
        rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0)
        torch.cuda.synchronize()
        runs.append((rew.copy(), ps.copy(), pr.copy(), sim.state_tensor("cuda").clone().cpu().numpy().view(np.uint64)))
    (r0, s0, c0, q0), (r1, s1, c1, q1) = runs
    assert sim.counters()["status"].max() == 0
    assert r0.sum() >= 3                                                       # a round with successes in it
    differing = int(np.any(q0 != q1, axis=1).sum())
    assert np.array_equal(r0, r1) and np.array_equal(s0, s1) and np.array_equal(c0, c1) and differing == 0, (differing, int((r0 != r1).sum()))


# ------------------------------------------------------------------ arm-link collision hulls
ARM_MESHES = ("base", "shoulder", "upperarm", "forearm", "wrist1", "wrist2", "wrist3")


def _arm_contact_state(model):
    """Reset state of the pile scene with object 0 moved INTO the forearm link (5 cm above the middle of its axis): the only way the state
    can be resolved is a contact between the object and the forearm's collision hull."""
    names = model.names["body"]
    o = Oracle(model)
    o.reset(20, 1, False)
    o.forward()
    xp = o.body_xpos()
    mid = 0.5 * (xp[names.index("forearm_link")] + xp[names.index("wrist_1_link")])
    st = o.get_state()
    q = st["qpos"].copy()
    q[8:11] = mid + np.array([0.0, 0.0, 0.05])
    o.set_state(qpos=q, qvel=st["qvel"], warmstart=st["warmstart"], pid=st["pid"])
    return o, dict(qpos=q[None], qvel=st["qvel"][None], warmstart=st["warmstart"][None], pid=st["pid"][None])


def _check_arm_hull_contact(model, sim, tol=(1e-9, 1e-9, 1e-9, 1e-7)):
    """tol = (contact point, normal, depth, relative qacc)"""
    arm_geoms = {g for g in range(model.ngeom) if model.geom_meshid[g] >= 0 and model.names["mesh"][model.geom_meshid[g]] in ARM_MESHES}
    assert len(arm_geoms) == 7 and all(model.geom_collide[g] for g in arm_geoms)
    o, state = _arm_contact_state(model)
    o.forward()
    sim.set_state(**state)
    d = sim.forward_debug()
    oc = o.contacts()
    on_arm = [c for c in oc if int(c[7]) in arm_geoms or int(c[8]) in arm_geoms]
    assert len(on_arm) >= 1 and min(c[0] for c in on_arm) < -0.01           # the forearm hull holds the object, centimetres deep
    assert d["ncon"][0] == len(oc)
    ec = d["contacts"][0][:len(oc)]
    for c in oc:
        best = min(ec, key=lambda e: np.abs(e[1:4] - c[1:4]).sum())
        assert np.abs(best[1:4] - c[1:4]).max() < tol[0] and np.abs(best[4:7] - c[4:7]).max() < tol[1] and abs(best[0] - c[0]) < tol[2], (best, c)
        assert {int(best[7]), int(best[8])} == {int(c[7]), int(c[8])}
    qacc = o.vec("qacc")
    assert np.abs(qacc[:6]).max() > 1.0                                      # the arm feels it
    assert np.abs(d["qacc"][0][:model.nv] - qacc).max() < tol[3] * max(1.0, np.abs(qacc).max())
    assert sim.counters()["status"][0] == 0


@pytest.fixture(scope="module")
def model_many_armcol():
    return load_model("many_objects_arm_collision")


def test_arm_link_hulls_collide_in_the_many_object_engine(model_many_armcol, emul_lib):
    """The reference collides the seven UR5 arm meshes (UR5gripper_2_finger_many_objects.xml:158-185). The many-object engine has a contact
    slot for every robot weld group, so the pile scene compiled with those hulls runs on it: an object pushed into the forearm produces the
    oracle's contact set and constrained acceleration, and the first steps of the drop agree as for the shipped scene."""
    m = model_many_armcol
    assert len(m.pair_geom1) > len(load_model(MANY).pair_geom1)
    sim = BatchSim(m, 1, lib_path=emul_lib)
    assert sim.variant == 1
    _check_arm_hull_contact(m, sim)
    sim.reset([20], 1, 0.0)
    _drop_parity(m, sim, 0, 20, 25, 1e-9)


def test_small_engine_rejects_arm_hulls_loudly(emul_lib):
    """The wavefront-per-scene engine keeps four robot contact slots (gripper only: 8 scenes per CU); a small scene compiled with the arm hulls is
    refused with a message, not silently simulated without them."""
    mjcf = pytest.importorskip("mujoco_rl_ur5_amd.mjcf")
    import os
    src = "/root/reference/UR5+gripper/UR5gripper_2_finger.xml"
    if not os.path.exists(src):
        pytest.skip("needs the reference MJCF")
    m = mjcf.compile_mjcf(src, arm_collision=True)
    with pytest.raises(Exception, match="robot weld groups|moving collidable geoms"):
        BatchSim(m, 1, lib_path=emul_lib)


@pytest.mark.gpu
def test_arm_link_hulls_on_gpu(model_many_armcol):
    sim = BatchSim(model_many_armcol, 2)
    assert sim.variant == 1
    # Round 6: the bounds of the shipped scene's forward parity (_forward_parity_from_engine_state: 1e-8 m / 1e-6 / 1e-10 m, qacc 1e-6 relative). Round 5 accepted 2e-5 / 1e-4 /
    # 1e-5 / 5e-3 here with a comment from round 2 ("the GPU build runs MPR in fused arithmetic"); the pile unit has compiled its geometry without fused multiply-adds
    # since round 3, and what is left between the two sides of a ROBOT contact is the arm's kinematics (Rodrigues + pointer jumping against the oracle's quaternion
    # chain: 1e-16 in the hull's pose).
    _check_arm_hull_contact(model_many_armcol, sim, tol=(1e-8, 1e-6, 1e-10, 1e-6))
    sim.reset(20 + np.arange(2, dtype=np.uint64), 1, 0.0)
    _drop_parity(model_many_armcol, sim, 1, 21, 25, 1e-9)


@pytest.mark.gpu
def test_every_handle_reads_its_own_model(model_many):
    """Round-4 verdict item 7 ("per-handle model behind a pointer"): rounds 1-4 kept ONE __constant__ model per engine unit and device, re-uploaded (43 KB behind a
    device synchronisation) whenever handles with different models took turns. Since round 5 the kernels read the model through the handle's own device copy
    (a kernel argument, struct Engine's only member: csrc/ur5_engine.h): ur5_create uploads it once -- counted by the test hook ur5_model_uploads -- and no launch ever uploads again, however
    the handles of a process alternate; and each handle's results are those it produces alone."""
    from mujoco_rl_ur5_amd.model import load_model
    m6, m1 = load_model("/UR5+gripper/UR5gripper_2_finger.xml"), load_model("it1_4box")
    alone = BatchSim(m6, 4)
    alone.reset(20 + np.arange(4, dtype=np.uint64), 1, 0.0)
    alone.step(40)
    q_alone = alone.get_state()["qpos"].copy()
    alone.close()
    six, it1 = BatchSim(m6, 4), BatchSim(m1, 2)                               # two DIFFERENT models of the small-scene unit
    u_small = six.model_uploads()
    pile = BatchSim(model_many, 2)
    u_pile = pile.model_uploads()
    six.reset(20 + np.arange(4, dtype=np.uint64), 1, 0.0)
    it1.reset(40 + np.arange(2, dtype=np.uint64), 1, 0.0)
    pile.reset(20 + np.arange(2, dtype=np.uint64), 1, 0.0)
    for _ in range(10):                                                       # alternating launches of the three handles
        six.step(4); it1.step(4); pile.step(2)
    assert six.model_uploads() == u_small and it1.model_uploads() == u_small and pile.model_uploads() == u_pile      # nothing re-uploaded
    assert np.array_equal(six.get_state()["qpos"], q_alone)                        # every bit of qpos as if the handle had the GPU to itself
    for s in (six, it1, pile):
        assert s.counters()["status"].max() == 0
    twin = BatchSim(m6, 2)                                                    # one upload per handle, at creation
    assert twin.model_uploads() == u_small + 1 and pile.model_uploads() == u_pile
