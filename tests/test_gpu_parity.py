"""`-m gpu`: the real libur5sim.so on an MI355X, through the C ABI, against the CPU oracle and the committed fixtures."""
import json
import os

import numpy as np
import pytest

from conftest import aimed_actions

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def native_mod():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    from mujoco_rl_ur5_amd import native
    native.load()          # raises loudly if csrc/libur5sim.so is missing: no silent fallback
    return native


def test_forward_quantities_on_gpu(native_mod, model_it1):
    from oracle.oracle import Oracle
    m = model_it1
    rng = np.random.default_rng(0)
    q = m.qpos0.copy()
    q[:8] = [0.3, -1.2, 1.1, -0.7, -1.0, 0.4, 0.2, 0.2]
    for k in range(4):
        qa = 8 + 7 * k
        quat = rng.normal(size=4)
        q[qa + 3:qa + 7] = quat / np.linalg.norm(quat)
        q[qa:qa + 3] = rng.uniform(-0.1, 0.1, size=3)
    v = rng.normal(size=m.nv) * 0.3
    ctrl = np.array([0.5, -1, 0.3, 0.2, -0.1, 0.7, -0.4])
    sim = native_mod.BatchSim(m, 4)
    o = Oracle(m)
    o.set_state(qpos=q, qvel=v); o.set_ctrl(ctrl)
    sim.set_state(qpos=q, qvel=v, warmstart=np.zeros(m.nv)); sim.set_ctrl(ctrl)
    o.forward()
    d = sim.forward_debug()
    assert np.abs(d["Mr"][0] - o.mass_matrix()[:8, :8]).max() < 1e-11
    assert np.abs(d["qacc_smooth"][0][:m.nv] - o.vec("qacc_smooth")).max() < 1e-7
    assert np.abs(d["qacc"][3][:m.nv] - o.vec("qacc")).max() < 1e-7


def test_grasp_attempt_matches_oracle_and_golden(native_mod, model_it1):
    """north_star bar: joint trajectories within 1e-4 rel, binary grasp success bit-exact."""
    from oracle.oracle import Oracle
    with open(os.path.join(GOLD, "oracle_grasp.json")) as f:
        gold = json.load(f)
    m = model_it1
    n = 16
    seeds = 20 + np.arange(n, dtype=np.uint64)
    sim = native_mod.BatchSim(m, n)
    sim.reset(seeds, 1, 1000.0)
    st = sim.get_state()
    for g in gold:
        e = g["seed"] - 20
        assert np.abs(st["qpos"][e] - np.array(g["settled_qpos"])).max() < 1e-9
    acts = aimed_actions(st["qpos"], 4)
    rots = (np.arange(n) // 4) % 6
    for g in gold:                      # golden runs used rot = env index % 6
        rots[g["seed"] - 20] = g["rot"]
    rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0)
    s2 = sim.get_state()
    for g in gold:
        e = g["seed"] - 20
        assert np.allclose(acts[e], g["action"], atol=1e-9)
        assert rew[e] == g["reward"] and ps[e].tolist() == g["phase_steps"] and pr[e].tolist() == g["phase_result"]
        assert np.abs(s2["qpos"][e][:8] - np.array(g["final_qpos"])[:8]).max() < 1e-6
    agree = 0
    for e in range(8):
        o = Oracle(m)
        o.reset(int(seeds[e]), 1, True)
        r, pso, pro = o.grasp_attempt(acts[e], int(rots[e]), 0)
        so = o.get_state()
        assert r == rew[e]                                                     # grasp bit
        assert pso.tolist() == ps[e].tolist()
        rel = np.abs(s2["qpos"][e][:8] - so["qpos"][:8]).max() / max(1.0, np.abs(so["qpos"][:8]).max())
        assert rel < 1e-9                                                      # measured 4e-16 (profiles/r04_p_drop_parity.log)
        # the objects are joints too. Scene e aims at box e % 4: the three boxes it never touches must agree like the arm does; the aimed
        # one is carried over the drop bin and released 0.5 m above its floor (check_mode 0). Measured on the oracle (round 4,
        # tests/test_oracle_kat.py::test_how_far_a_small_scene_attempt_amplifies_a_perturbation): a 1e-13 m perturbation stays below 1e-9 -- summation order
        # is NOT what separates kernel and oracle here --, a 1e-6 m one (the scale at which two MPR runs on the finger hulls differ: its tolerance, fused
        # arithmetic in the kernel) moves the released box by up to 0.11 m while reward and all step counts stay equal
        eo = np.abs(s2["qpos"][e][8:] - so["qpos"][8:]).reshape(-1, 7)[:, :3].max(axis=1)
        others = np.delete(eo, e % 4)
        # Round 4 measured all of it (tools/gpu_drop_parity.py, 32 scenes, 22 successes: arm 4e-16, untouched boxes <= 3e-11, the released box <= 5e-11) and tightened the
        # bounds that dated from the 32-vertex hull approximations: 5 cm -> 1e-7 m for the released box, 1e-6 -> 1e-8 for the others
        assert others.max() < 1e-8, (e, eo)
        assert eo[e % 4] < 1e-7, (e, eo)
        agree += 1
    assert agree == 8 and set(np.unique(rew)) == {0, 1}
    assert np.all(sim.counters()["status"] == 0)


def test_batch_position_and_rerun_determinism(native_mod, model_it1):
    m = model_it1
    a = native_mod.BatchSim(m, 64)
    b = native_mod.BatchSim(m, 1)
    seeds = 100 + np.arange(64, dtype=np.uint64)
    a.reset(seeds, 1, 400.0)
    b.reset(seeds[37:38], 1, 400.0)
    assert np.array_equal(a.get_state()["qpos"][37], b.get_state()["qpos"][0])   # scene 37 of 64 == the same scene alone
    first = a.get_state()["qpos"].copy()
    a.reset(seeds, 1, 400.0)
    # controller state persists across resets (reference semantics), so compare through a fresh handle instead
    c = native_mod.BatchSim(m, 64)
    c.reset(seeds, 1, 400.0)
    assert np.array_equal(c.get_state()["qpos"], first)


def test_full_size_batch_properties(native_mod, model_it1):
    """BASELINE.json config 2 size (4096 scenes): size-independent properties instead of a 4096-scene oracle run."""
    m = model_it1
    n = 4096
    sim = native_mod.BatchSim(m, n)
    seeds = 20 + np.arange(n, dtype=np.uint64)
    sim.reset(seeds, 1, 1000.0)
    st = sim.get_state()
    assert np.isfinite(st["qpos"]).all() and np.isfinite(st["qvel"]).all()
    quats = st["qpos"][:, 8:].reshape(n, 4, 7)[:, :, 3:]
    assert np.abs(np.linalg.norm(quats, axis=2) - 1).max() < 1e-12
    z = 0.95 + 0.1 * np.arange(4) + st["qpos"][:, 8:].reshape(n, 4, 7)[:, :, 2]
    on_table = np.abs(z - 0.931) < 2e-3
    stacked = np.abs(z - 0.972) < 4e-3                 # a box that landed on another one
    assert (on_table | stacked | (z < 0.93)).mean() > 0.98
    assert np.abs(st["qpos"][:, 6] - st["qpos"][:, 7]).max() < 1e-2           # `fingers` equality holds (xml :333)
    c = sim.counters()
    assert np.all(c["status"] == 0) and np.all(c["total_steps"] == c["total_steps"][0])
    acts = aimed_actions(st["qpos"], 4)
    rew, ps, pr = sim.grasp_attempt(acts, rot=np.arange(n) % 6, check_mode=1)
    assert set(np.unique(rew)) <= {0, 1} and 0.05 < rew.mean() < 0.98
    q_end = sim.get_state()["qpos"]
    assert np.isfinite(q_end).all()
    # ... and the oracle itself on a sample of the batch (round-5 verdict 11): 64 of the 4096 scenes, reset + settle + attempt on the host, against the scene's place in the
    # full-size launch -- reward bit, the 12 phase step counts, arm joints, untouched objects
    from oracle.oracle import Oracle
    for e in np.random.default_rng(11).choice(n, 64, replace=False):
        o = Oracle(m)
        o.reset(int(seeds[e]), 1, True)
        assert np.array_equal(o.get_state()["qpos"][:8], st["qpos"][e][:8]) or np.abs(o.get_state()["qpos"] - st["qpos"][e]).max() < 1e-8
        q_before = o.get_state()["qpos"].copy()
        r, pso, pro = o.grasp_attempt(acts[e], int(e % 6), 1)
        assert r == rew[e] and pso.tolist() == ps[e].tolist() and pro.tolist() == pr[e].tolist(), (e, r, rew[e], pso, ps[e])
        assert np.abs(q_end[e][:8] - o.get_state()["qpos"][:8]).max() < 1e-6, e
        untouched_objects_agree(q_end[e], q_before, o.get_state()["qpos"])


def untouched_objects_agree(q_gpu, q_before, q_after, tol=1e-8):
    """north_star's "joint trajectories" include the OBJECT joints (round-5 verdict 1d). Objects the attempt never touched -- the oracle moved none of their seven
    coordinates by more than 1e-6 -- are where they were on both sides, to `tol` (metres / quaternion components); returns how many objects that was. (An object the
    gripper released follows another trajectory on every rounding: it is bounded where the step counts are equal, test_grasp_bit_agreement_statistics.)"""
    ob, oa, og = q_before[8:].reshape(-1, 7), q_after[8:].reshape(-1, 7), q_gpu[8:].reshape(-1, 7)
    still = np.abs(oa - ob).max(axis=1) < 1e-6
    assert np.abs(og[still] - oa[still]).max(initial=0.0) < tol, (np.abs(og - oa).max(axis=1), still)
    return int(still.sum())


def test_six_object_scene_nv44_kernel(native_mod, model_2f):
    """The in-tree UR5gripper_2_finger.xml (3 boxes + 3 spheres, nv = 44) runs through the NV=44 instantiation: settle + one
    grasp attempt against the oracle."""
    from oracle.oracle import Oracle
    m = model_2f
    sim = native_mod.BatchSim(m, 4)
    sim.reset(20 + np.arange(4, dtype=np.uint64), 1, 1000.0)
    o = Oracle(m)
    o.reset(21, 1, True)
    st = sim.get_state()
    assert np.abs(st["qpos"][1] - o.get_state()["qpos"]).max() < 1e-8
    acts = aimed_actions(st["qpos"], 6)
    rew, ps, pr = sim.grasp_attempt(acts, rot=0, check_mode=0)
    q_before = o.get_state()["qpos"].copy()
    r, pso, pro = o.grasp_attempt(acts[1], 0, 0)
    assert r == rew[1] and pso.tolist() == ps[1].tolist()
    assert np.abs(sim.get_state()["qpos"][1][:8] - o.get_state()["qpos"][:8]).max() < 1e-6
    assert untouched_objects_agree(sim.get_state()["qpos"][1], q_before, o.get_state()["qpos"]) >= 3      # one of six is aimed at; at least three never move
    assert np.all(sim.counters()["status"] == 0)


def test_grasp_bit_agreement_statistics(native_mod, model_it1):
    """agreement rate of the binary grasp outcome over many scenes, all six wrist rotations, a third of the
    attempts aimed off-centre (profiles/r01_e_grasp_agreement_384scenes.json has the 384-scene run: 100 %)."""
    from oracle.oracle import Oracle
    m = model_it1
    n = 48
    sim = native_mod.BatchSim(m, n)
    seeds = 5000 + np.arange(n, dtype=np.uint64)
    sim.reset(seeds, 1, 1000.0)
    acts = aimed_actions(sim.get_state()["qpos"], 4)
    rng = np.random.default_rng(3)
    acts[::3, :2] += rng.uniform(-0.012, 0.012, size=(len(acts[::3]), 2))
    rots = np.arange(n) % 6
    rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=1)
    s2 = sim.get_state()["qpos"]
    agree, worst, worst_obj = 0, 0.0, 0.0
    for e in range(n):
        o = Oracle(m)
        o.reset(int(seeds[e]), 1, True)
        r, pso, pro = o.grasp_attempt(acts[e], int(rots[e]), 1)
        agree += int(r == rew[e])
        so = o.get_state()["qpos"]
        worst = max(worst, np.abs(s2[e][:8] - so[:8]).max() / max(1.0, np.abs(so[:8]).max()))
        if pso.tolist() == ps[e].tolist():
            worst_obj = max(worst_obj, np.abs(s2[e][8:] - so[8:]).max())
    assert agree == n                    # bit-exact grasp outcomes
    assert worst < 1e-4                  # north_star: joint trajectories within 1e-4 rel
    assert worst_obj < 1e-5, worst_obj   # ... the object joints included (IT1: the grasped box is still in the gripper / on the plate at the end)
    assert 0.1 < rew.mean() < 0.95


def test_device_pointer_entry(native_mod, model_it1):
    import torch
    m = model_it1
    sim = native_mod.BatchSim(m, 8)
    sim.reset(20 + np.arange(8, dtype=np.uint64), 1, 1000.0)
    acts = aimed_actions(sim.get_state()["qpos"], 4)
    a = torch.zeros((8, 8), dtype=torch.float64, device="cuda")
    a[:, :3] = torch.from_numpy(acts).cuda()
    r = torch.full((8,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    sim.grasp_attempt_dev(a.data_ptr(), r.data_ptr(), check_mode=0)
    sim.sync()
    assert sim.last_launch_ms() > 0
    rew_dev = r.cpu().numpy()
    sim2 = native_mod.BatchSim(m, 8)
    sim2.reset(20 + np.arange(8, dtype=np.uint64), 1, 1000.0)
    rew, _, _ = sim2.grasp_attempt(acts, rot=0, check_mode=0)
    assert rew_dev.tolist() == rew.tolist()


def random_agent_attempts(model, n, seed=0):
    """The reference's random agent (example_agent.py: env.action_space.sample()): a pixel drawn uniformly from the WHOLE 200x200 image and one of
    the six rotations, back-projected at the fixed table height (GraspEnv.step, GraspingEnv.py:100-104). Most pixels show floor, bins or the
    pedestal: these attempts take the script's failure exits -- no IK solution ("No valid joint angles received", MujocoController.py:505-517),
    a descent stopped by a wall or a box ("max. steps reached", GraspingEnv.py:241-248) -- which aimed grasps never reach."""
    from mujoco_rl_ur5_amd.controller import MJ_Controller

    class _NoSim:
        n = 1
    cam = MJ_Controller(model, simulation=_NoSim())
    rng = np.random.default_rng(seed)
    px, py, rots = rng.integers(0, 200, n), rng.integers(0, 200, n), rng.integers(0, 6, n)
    acts = np.array([[*cam.pixel_2_world(int(x), int(y), 2.0 - 0.91)[:2], 0.91] for x, y in zip(px, py)])
    return acts, rots


def check_random_agent_parity(BatchSim, model, n, **kw):
    from oracle.oracle import Oracle
    acts, rots = random_agent_attempts(model, n)
    seeds = 20 + np.arange(n, dtype=np.uint64)
    sim = BatchSim(model, n, **kw)
    sim.reset(seeds, 1, 1000.0)
    rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0)
    q = sim.get_state()["qpos"]
    codes, still = set(), 0
    for e in range(n):
        o = Oracle(model)
        o.reset(int(seeds[e]), 1, True)
        q_before = o.get_state()["qpos"].copy()
        r, pso, pro = o.grasp_attempt(acts[e], int(rots[e]), 0)
        assert r == rew[e] and pro.tolist() == pr[e].tolist(), (e, acts[e], pro, pr[e])
        if pro[3] == 1:
            # A blocked descent (GraspingEnv.py:241-248) pushes for 300 steps on whatever stopped it. In scene 35 of the six-object set the
            # gripper comes down on the bin's rim and levers a box over the wall: where the box tumbles to is decided at rounding level, and
            # oracle, lane emulation and GPU (three summation orders) then need 252 / 255 / 253 steps for the retreat that follows while it
            # is still rolling. Exits, reward and the phases up to the blockage stay equal; the later counts are bounded, not equal.
            assert pso[:4].tolist() == ps[e][:4].tolist() and np.abs(pso - ps[e]).max() <= 8, (e, acts[e], pso, ps[e])
            assert np.abs(q[e][:8] - o.qpos[:8]).max() < 1e-2, e
            codes.add(tuple(pro.tolist()))
            continue
        assert pso.tolist() == ps[e].tolist(), (e, acts[e], pso, ps[e])
        assert np.abs(q[e][:8] - o.qpos[:8]).max() < 1e-6, e
        still += untouched_objects_agree(q[e], q_before, o.get_state()["qpos"])
        codes.add(tuple(pro.tolist()))
    assert sim.counters()["status"].max() == 0 and still >= 2 * n                  # most objects of a random attempt are never touched: they are bounded, not skipped
    return codes


def test_random_agent_attempts_take_the_same_exits_as_the_oracle(native_mod, model_it1):
    codes = check_random_agent_parity(native_mod.BatchSim, model_it1, 96)
    assert any(c[3] == 2 for c in codes) and any(c[3] == 1 for c in codes) and any(c[3] == 0 for c in codes)   # IK failure, blocked descent, plain


def test_random_agent_attempts_on_the_six_object_scene(native_mod, model_2f):
    """The same on the in-tree scene (3 boxes + 3 spheres, the 44-dof kernel instantiation)."""
    codes = check_random_agent_parity(native_mod.BatchSim, model_2f, 64)
    assert any(c[3] == 2 for c in codes) and any(c[3] == 0 for c in codes)


def test_ik_values_and_move_ee_on_gpu(native_mod, model_it1):
    """ur5_ik / ur5_move_ee directly: a grid of targets over the workspace, some unreachable; joint solutions equal the oracle's
    to 1e-9, the same targets fail, and move_ee from the home pose takes the same number of steps to the same pose."""
    from oracle.oracle import Oracle
    m = model_it1
    xs, ys, zs = np.meshgrid(np.linspace(-0.35, 0.65, 6), np.linspace(-0.9, 0.15, 6), [0.95, 1.1, 1.3])
    targets = np.stack([xs.ravel(), ys.ravel(), zs.ravel()], axis=1)
    targets = np.concatenate([targets, [[2.0, 2.0, 2.0], [0.0, 0.0, 3.0], [0.0, -0.6, 0.2]]])
    n = len(targets)
    sim = native_mod.BatchSim(m, n)
    sim.reset(20 + np.arange(n, dtype=np.uint64), 1, 0.0)
    q5, ok = sim.ik(targets)
    o = Oracle(m)
    o.reset(20, 1, False)
    nfail = 0
    for e in range(n):
        oko, q5o = o.ik(targets[e])
        assert bool(ok[e] == 0) == oko, (e, targets[e], ok[e], oko)
        if oko:
            assert np.abs(q5[e] - q5o).max() < 1e-9, (e, targets[e])
        nfail += not oko
    assert 3 <= nfail < n // 2
    res, steps = sim.move_ee(targets, 0.02, 1500)
    q = sim.get_state()["qpos"]
    for e in range(0, n, 7):
        o = Oracle(m)
        o.reset(20 + e, 1, False)
        r, s_ = o.move_ee(targets[e], 0.02, 1500)
        assert r == res[e] and s_ == steps[e], (e, targets[e], r, res[e], s_, steps[e])
        assert np.abs(q[e][:8] - o.qpos[:8]).max() < 1e-7
