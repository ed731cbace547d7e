"""The engine's DEVICE code path on the CPU: csrc/ur5_engine.h compiled without -DUR5_EMUL, its HIP intrinsics emulated with the semantics of one
64-lane wavefront (tests/emul/ur5_simt_shim.h: a fibre per lane, a rendezvous per cross-lane instruction). The plain lane emulation of
test_engine_emul.py needs LDS stand-ins for what the GPU keeps in registers; this build runs the code that ships -- DPP reductions, v_readlane
pivots of the row factorisation, the block-parallel register Cholesky of the robot factors and of the block-diagonal Newton path, ballot
compactions, the 8-lanes-per-pair MPR with its DPP vertex exchange -- against the oracle, without a GPU."""
import numpy as np
import pytest

from conftest import aimed_actions
from mujoco_rl_ur5_amd.native import BatchSim
from oracle.oracle import Oracle
from test_engine_emul import _random_state


def test_forward_quantities_through_the_device_path(model_it1, simt_lib):
    m = model_it1
    sim = BatchSim(m, 2, lib_path=simt_lib)
    q, v = _random_state(m, 0)
    ctrl = np.array([0.5, -1, 0.3, 0.2, -0.1, 0.7, -0.4])
    o = Oracle(m)
    o.set_state(qpos=q, qvel=v)
    o.set_ctrl(ctrl)
    sim.set_state(qpos=q, qvel=v, warmstart=np.zeros(m.nv))
    sim.set_ctrl(ctrl)
    o.forward()
    d = sim.forward_debug()
    assert np.abs(d["Mr"][1] - o.mass_matrix()[:8, :8]).max() < 1e-12
    assert np.abs(d["qacc_smooth"][1][:m.nv] - o.vec("qacc_smooth")).max() < 1e-8    # robot LDL^T in registers (blk_cholesky / blk_solve)
    assert np.abs(d["qacc"][1][:m.nv] - o.vec("qacc")).max() < 1e-8


def test_settle_and_grasp_attempts_equal_the_oracle(model_it1, simt_lib):
    """reset_model + two whole move_and_grasp scripts (GraspingEnv.py:205-386): ~5 000 physics steps through every phase of the device code --
    free flight, boxes on the plate (block-diagonal Newton in registers), the gripper's hulls on a box (cooperative MPR, coupled Hessian with
    v_readlane pivots), IK, PID, the script interpreter."""
    m = model_it1
    n = 2
    seeds = np.array([20, 23], dtype=np.uint64)
    sim = BatchSim(m, n, lib_path=simt_lib)
    sim.reset(seeds, 1, 1000.0)
    st = sim.get_state()
    acts = aimed_actions(st["qpos"], 4)
    acts[1, :2] += [0.008, -0.006]                                               # off-centre: the fingers push the box before they close
    rots = np.array([0, 4])
    rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=1)
    s2 = sim.get_state()
    for e in range(n):
        o = Oracle(m)
        o.reset(int(seeds[e]), 1, True)
        assert np.abs(st["qpos"][e] - o.get_state()["qpos"]).max() < 1e-12       # settled state
        r, pso, pro = o.grasp_attempt(acts[e], int(rots[e]), 1)
        assert r == rew[e] and pso.tolist() == ps[e].tolist() and pro.tolist() == pr[e].tolist(), (e, pso, ps[e])
        assert np.abs(s2["qpos"][e][:8] - o.qpos[:8]).max() < 1e-9 and np.abs(s2["qpos"][e][8:] - o.qpos[8:]).max() < 1e-7, e
    assert sim.counters()["status"].max() == 0 and sim.counters()["ncon_max"].max() >= 18


def test_six_object_scene_and_failure_exits(model_2f, simt_lib):
    """The NV = 44 instantiation (3 boxes + 3 spheres): settle, then the reference's random agent at pixels that take the script's IK-failure exit
    and a normal attempt."""
    from test_gpu_parity import random_agent_attempts
    m = model_2f
    acts, rots = random_agent_attempts(m, 16)
    o_codes = {}
    for e in range(16):                                                          # pick one IK failure and one plain attempt with the oracle (fast)
        o = Oracle(m)
        o.reset(20 + e, 1, True)
        r, pso, pro = o.grasp_attempt(acts[e], int(rots[e]), 0)
        o_codes.setdefault(int(pro[3]), (e, r, pso, pro, o.qpos.copy()))
    picks = [o_codes[k] for k in (2, 0) if k in o_codes]
    assert len(picks) == 2, sorted(o_codes)
    idx = [p[0] for p in picks]
    sim = BatchSim(m, 2, lib_path=simt_lib)
    sim.reset(20 + np.array(idx, dtype=np.uint64), 1, 1000.0)
    rew, ps, pr = sim.grasp_attempt(acts[idx], rot=rots[idx], check_mode=0)
    q = sim.get_state()["qpos"]
    for k, (e, r, pso, pro, qo) in enumerate(picks):
        assert r == rew[k] and pro.tolist() == pr[k].tolist() and pso.tolist() == ps[k].tolist(), (e, pso, ps[k])
        assert np.abs(q[k][:8] - qo[:8]).max() < 1e-8, e
    assert sim.counters()["status"].max() == 0


def test_six_object_grasp_holds_through_the_device_code_s_coupled_hessian(model_2f, simt_lib):
    """The DEVICE code path of the NV = 44 kernel with a held object: rows of the packed Hessian into the lanes' registers, factorisation by lane broadcasts, the
    factor transposed through the Hessian's LDS -- the same LDS whose tail holds the twists, the search direction's images, M search and the aref offsets in this
    instantiation (Lds::TAIL). One grasp that holds and one that loses the object: reward, phase step counts, result codes equal the oracle's."""
    from conftest import aimed_actions
    m = model_2f
    cases = [(20, 0, 0), (20, 2, 3)]
    sim = BatchSim(m, len(cases), lib_path=simt_lib)
    sim.reset(np.array([c[0] for c in cases], dtype=np.uint64), 1, 1000.0)
    q0 = sim.get_state()["qpos"]
    acts = np.array([aimed_actions(q0[e][None], 6, first_id=c[1])[0] for e, c in enumerate(cases)])
    rew, ps, pr = sim.grasp_attempt(acts, rot=[c[2] for c in cases], check_mode=0)
    q = sim.get_state()["qpos"]
    for e, (seed, k, rot) in enumerate(cases):
        o = Oracle(m)
        o.reset(seed, 1, True)
        r, pso, pro = o.grasp_attempt(acts[e], rot, 0)
        assert r == rew[e] and pso.tolist() == ps[e].tolist() and pro.tolist() == pr[e].tolist(), (e, pso, ps[e])
        assert np.abs(q[e][:8] - o.qpos[:8]).max() < 1e-8 and np.abs(q[e] - o.qpos).max() < 1e-6, e
    assert rew.tolist() == [1, 0] and sim.counters()["status"].max() == 0


def test_ik_and_move_ee_through_the_device_path(model_it1, simt_lib):
    m = model_it1
    targets = np.array([[0.0, -0.6, 1.1], [0.2, -0.45, 0.95], [0.6, 0.1, 1.2], [2.0, 2.0, 2.0]])
    sim = BatchSim(m, len(targets), lib_path=simt_lib)
    sim.reset(20 + np.arange(len(targets), dtype=np.uint64), 1, 0.0)
    q5, ok = sim.ik(targets)
    o = Oracle(m)
    o.reset(20, 1, False)
    for e, t in enumerate(targets):
        oko, q5o = o.ik(t)
        assert bool(ok[e] == 0) == oko
        if oko:
            assert np.abs(q5[e] - q5o).max() < 1e-10


# ------------------------------------------------------------------ the many-object kernel: 256 fibres = 4 wavefronts per scene
def test_many_object_device_path(simt_lib):
    """ur5m_run_kernel's own code -- cross-wave reductions through LDS, the envelope Cholesky with one wavefront per panel of a level, factor
    reuse, __syncthreads between 4 waves -- against the oracle: the first contacts of the 40-object drop step by step, then contacts and
    constrained acceleration in the dense pile 0.4 s later, then an object pushed into the forearm's collision hull (arm-collision asset)."""
    import test_many_objects as T
    from mujoco_rl_ur5_amd.model import load_model
    m = load_model(T.MANY)
    sim = BatchSim(m, 1, lib_path=simt_lib)
    assert sim.variant == 1
    sim.reset([20], 1, 0.0)
    T._drop_parity(m, sim, 0, 20, 40, 1e-9)
    sim.reset([21], 1, 0.0)
    sim.step(200)
    st = sim.get_state()
    o = Oracle(m)
    o.set_state(qpos=st["qpos"][0], qvel=st["qvel"][0], warmstart=st["warmstart"][0], pid=st["pid"][0])
    o.forward()
    d = sim.forward_debug()
    oc = o.contacts()
    assert d["ncon"][0] == len(oc) and len(oc) >= 15
    ec = d["contacts"][0][:len(oc)]
    for c in oc:
        best = min(ec, key=lambda e: np.abs(e[1:4] - c[1:4]).sum())
        assert np.abs(best[1:4] - c[1:4]).max() < 1e-9 and np.abs(best[4:7] - c[4:7]).max() < 1e-9 and abs(best[0] - c[0]) < 1e-9
    qacc = o.vec("qacc")
    assert np.abs(d["qacc"][0][:m.nv] - qacc).max() < 1e-6 * max(1.0, np.abs(qacc).max())
    assert sim.counters()["status"][0] == 0
    ma = load_model("many_objects_arm_collision")
    T._check_arm_hull_contact(ma, BatchSim(ma, 1, lib_path=simt_lib))


def test_cross_lane_operation_budget_of_a_step(model_it1, simt_lib):
    """What one physics step costs in cross-lane instructions (lane 0's counts in the SIMT build): a guard against an accidental extra barrier or
    reduction in the hot loop, and the numbers DESIGN.md quotes. Settled 4-box scene: ~46 wave barriers, ~200 DPP moves (a 64-lane sum of a
    double is 12 of them + 2 v_readlane), ~30 v_readlane, ~120 shuffles (the block-parallel register Cholesky), 6 ballots."""
    import ctypes as C
    sim = BatchSim(model_it1, 1, lib_path=simt_lib)

    def counts():
        out = (C.c_long * 8)()
        sim.lib.ur5_simt_op_counts(out)
        return np.array(list(out), dtype=float)
    sim.reset([20], 1, 1000.0)
    c0 = counts()
    sim.step(50)
    dpp, readlane, shuffle, ballot, wave_barrier, block_barrier, atomic, _ = (counts() - c0) / 50
    assert 30 <= wave_barrier <= 60 and block_barrier == 0, wave_barrier
    assert 110 <= dpp <= 260 and 20 <= readlane <= 60 and 80 <= shuffle <= 160 and ballot <= 8, (dpp, readlane, shuffle, ballot)   # 144 DPP moves since the line search reuses its last cost (round 4)
    assert 10 <= atomic <= 40, atomic


def test_results_do_not_depend_on_the_lane_schedule(model_it1, simt_lib):
    """Race detector. The fibres of a wavefront are normally scheduled in ascending lane order; here every scheduler pass uses a fresh permutation
    (and, for the pile, descending order). The engine may only rely on what its barriers and cross-lane instructions guarantee, so the outcome must
    stay the same (for the wavefront-per-scene kernel: up to the order in which its LDS atomics land, rounding level): a missing SYNC between one lane's LDS write and another lane's
    read would change the result with the schedule -- or surface the NaN poison the LDS image starts with."""
    import test_many_objects as T
    from mujoco_rl_ur5_amd.model import load_model
    m = model_it1
    sim = BatchSim(m, 1, lib_path=simt_lib)
    try:
        sim.lib.ur5_simt_set_order(2)
        sim.reset([20], 1, 1000.0)
        st = sim.get_state()
        acts = aimed_actions(st["qpos"], 4)
        rew, ps, pr = sim.grasp_attempt(acts, rot=0, check_mode=1)
        o = Oracle(m)
        o.reset(20, 1, True)
        assert np.abs(st["qpos"][0] - o.get_state()["qpos"]).max() < 1e-12
        r, pso, pro = o.grasp_attempt(acts[0], 0, 1)
        assert r == rew[0] and pso.tolist() == ps[0].tolist() and pro.tolist() == pr[0].tolist()
        assert np.abs(sim.get_state()["qpos"][0] - o.qpos).max() < 1e-9 and sim.counters()["status"][0] == 0
        mm = load_model(T.MANY)
        for order in (1, 2):
            sim.lib.ur5_simt_set_order(order)
            big = BatchSim(mm, 1, lib_path=simt_lib)
            big.reset([20], 1, 0.0)
            T._drop_parity(mm, big, 0, 20, 30, 1e-9)
        # Round 4: the pile kernel no longer sums anything with LDS float atomics (contacts sorted into pair order, fixed-order gathers), so a dense pile
        # must come out BIT-IDENTICAL under ascending, descending and freshly permuted lane schedules of its four wavefronts -- the CPU twin of
        # tests/test_many_objects.py::test_pile_kernel_is_run_to_run_deterministic_on_gpu. (With the atomics the same 40 steps differed by 1e-13 .. 1e-4.)
        sim.lib.ur5_simt_set_order(0)
        big = BatchSim(mm, 1, lib_path=simt_lib)
        big.reset([21], 1, 0.0)
        big.step(200)
        s0 = big.get_state()
        assert len(big.forward_debug()["contacts"][0]) and big.forward_debug()["ncon"][0] >= 15
        outs = []
        for order in (0, 1, 2):
            sim.lib.ur5_simt_set_order(order)
            big.set_state(**s0)
            big.step(40)
            st = big.get_state()
            outs.append(np.concatenate([st["qpos"][0], st["qvel"][0], st["warmstart"][0]]))
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), (np.abs(outs[0] - outs[1]).max(), np.abs(outs[0] - outs[2]).max())
        assert big.counters()["status"][0] == 0
    finally:
        sim.lib.ur5_simt_set_order(0)


def test_the_race_detector_sees_a_planted_race(model_it1, simt_lib):
    """Self-test of the schedule-permutation detector: with ONE wave barrier of the launch ignored (ur5_simt_skip_barrier) the first 30 steps of
    a drop must stop matching the oracle. Over a window of barriers inside the first step there must be (a) barriers that ascending lane order
    happens to survive and a permuted order does not -- the reason the detector permutes --, and (b) barriers whose loss reads the NaN poison of
    the LDS image: the state guard then resets the scene and flags it (status bit 2). With no barrier skipped both orders match to rounding.
    (The window is scanned instead of naming barrier numbers: every SYNC added to the engine renumbers them.)"""
    m = model_it1
    o = Oracle(m)
    o.reset(20, 1, False)
    o.step(30)
    ref = o.get_state()["qpos"]
    sim = BatchSim(m, 1, lib_path=simt_lib)

    def run(skip, order):
        sim.lib.ur5_simt_skip_barrier(skip)
        sim.lib.ur5_simt_set_order(order)
        sim.reset([20], 1, 0.0)
        sim.step(30)
        err = np.abs(sim.get_state()["qpos"][0] - ref).max()
        return err, int(sim.counters()["status"][0])
    try:
        assert run(-1, 0)[0] < 1e-12 and run(-1, 2)[0] < 1e-12
        only_permuted, flagged = [], []
        for b in range(96, 112):
            (e0, s0), (e2, s2) = run(b, 0), run(b, 2)
            if e0 < 1e-12 and s0 == 0 and (e2 > 1e-10 or s2 != 0):
                only_permuted.append(b)
            if s0 & 2 or s2 & 2:
                flagged.append(b)
        assert only_permuted and flagged, (only_permuted, flagged)
    finally:
        sim.lib.ur5_simt_skip_barrier(-1)
        sim.lib.ur5_simt_set_order(0)
