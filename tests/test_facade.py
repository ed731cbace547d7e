"""GraspEnv / MJ_Controller keep the reference's Python surface; run on the lane emulation."""
import numpy as np
import pytest

from mujoco_rl_ur5_amd.controller import MJ_Controller, IK_FAIL_STRING
from mujoco_rl_ur5_amd.envs import GraspEnv, make


@pytest.fixture(scope="module")
def env(model_it1, emul_lib):
    return GraspEnv(file=model_it1, show_obs=False, render=False, n_envs=1, observation="flat", _lib_path=emul_lib)


def test_env_surface_matches_reference(env):
    assert env.rotations == {0: 0, 1: 30, 2: 60, 3: 90, 4: -30, 5: -60} and env.TABLE_HEIGHT == 0.91
    assert env.action_space.nvec.tolist() == [40000, 6]
    assert env.model.camera_name2id("top_down") == 1 and np.allclose(env.model.cam_pos0[1], [0, -0.6, 2.0])
    obs = env.reset()
    assert obs["rgb"].shape == (200, 200, 3) and obs["depth"].shape == (200, 200)
    assert np.allclose(obs["depth"], 2.0 - 0.91)


def test_random_agent_loop_like_example_agent(env):
    """example_agent.py:15-27 shape: reset, then action_space.sample() steps; rewards are 0/1, done stays False."""
    from mujoco_rl_ur5_amd.envs import MultiDiscrete
    env.action_space = MultiDiscrete(env.action_space.nvec, seed=3)          # deterministic: an unseeded sample can knock box 0 off the plate before it is aimed at
    env.reset()
    seen = []
    for _ in range(2):
        action = env.action_space.sample()
        obs, reward, done, info = env.step(action)
        assert reward in (0, 1) and done is False and obs["depth"].shape == (200, 200)
        seen.append(info["skipped"])
    # a pixel aimed at an object should be able to succeed
    qpos = env.sim.get_state()["qpos"][0]
    box = qpos[8:11]
    px = env.controller.world_2_pixel([box[0], -0.6 + box[1], 0.91])
    obs, reward, done, info = env.step([px[1] * 200 + px[0], 0])
    assert info["phase_steps"][0] > 0 and reward in (0, 1)


def test_skip_rule_leaves_the_scene_untouched(env):
    env.reset()
    before = env.sim.get_state()["qpos"].copy()
    obs, reward, done, info = env.step([0 * 200 + 100, 0])       # top image row -> world y = -0.6 + 100*1.09/241 > -0.3
    assert info["skipped"] and reward == 0
    assert np.array_equal(env.sim.get_state()["qpos"], before)


def test_controller_strings_and_groups(model_it1, emul_lib):
    c = MJ_Controller(model_it1, n_envs=1, _lib_path=emul_lib)
    assert dict(c.groups) == {"All": [0, 1, 2, 3, 4, 5, 6], "Arm": [0, 1, 2, 3, 4], "Gripper": [6]}
    assert c.actuated_joint_ids.tolist() == [0, 1, 2, 3, 4, 5, 6]
    assert np.allclose(c.current_target_joint_values, [0, -1.57, 1.57, -1.57, -1.57, 0, 0])
    r = c.move_group_to_joint_target(group="Arm", target=[0, -1.2, 1.2, -1.57, -1.57], tolerance=1e-9, max_steps=20, quiet=True)
    assert r == "max. steps reached: 20" and c.last_steps == 21
    assert c.move_ee([3.0, 3.0, 3.0], max_steps=10, tolerance=0.1) == IK_FAIL_STRING and c.last_steps == 0
    assert c.ik([3.0, 3.0, 3.0]) is None and len(c.ik([0.0, -0.6, 1.1])) == 5
    assert c.open_gripper(half=True, quiet=True) in ("success", "max. steps reached: 1000")
    c.actuate_joint_group("Gripper", [0.5])
    assert c.sim.get_ctrl()[0, 6] == 0.5
    c.stay(20)
    assert isinstance(c.grasp(quiet=True), bool)
    rgb, d = c.get_image_data(width=40, height=40)
    assert rgb.shape == (40, 40, 3) and rgb.dtype == np.uint8 and 0 < d.min() and d.max() <= 1.0


def test_batched_env_and_make(model_it1, emul_lib):
    e = make("gym_grasper:Grasper-v0", file=model_it1, n_envs=2, show_obs=False, observation="flat", _lib_path=emul_lib)
    obs = e.reset()
    assert obs["depth"].shape == (2, 200, 200)
    a = np.stack([e.action_space.sample(), e.action_space.sample()])
    obs, reward, done, info = e.step(a)
    assert reward.shape == (2,) and set(np.unique(reward)) <= {0, 1}


def test_default_env_is_the_many_object_scene(emul_lib):
    """GraspEnv() with no file argument is the reference's 40-object pile (GraspingEnv.py:28); libur5sim serves it with its
    many-object engine variant behind the same entry points."""
    e = GraspEnv(n_envs=1, show_obs=False, observation="flat", _lib_path=emul_lib)
    assert e.sim.variant == 1 and e.model.nv == 248 and e.sim.nq == 288
    obs = e.reset()
    assert obs["depth"].shape == (200, 200) and obs["rgb"].shape == (200, 200, 3)
    z = e.sim.get_state()["qpos"][0][8:].reshape(-1, 7)[:, 2]
    assert (z < 1.2).all() and (z > 0.6).all()                       # the pile has dropped into the bin (floor 0.89) or beside it
    c = e.sim.counters()
    assert c["status"][0] == 0 and 5 < c["ncon_max"][0] < 160
    r = e.controller.move_group_to_joint_target(group="Gripper", target=[0.2], tolerance=0.05, max_steps=200, quiet=True)
    assert r == "success"


def test_reset_dev_matches_host_reset_and_leaves_unflagged_scenes_alone(model_it1, emul_lib):
    """ur5_reset_dev (device-side GraspEnv.reset_model for flagged scenes) == ur5_reset scene by scene; unflagged scenes untouched."""
    import ctypes as C
    import numpy as np
    from mujoco_rl_ur5_amd.native import BatchSim
    n = 4
    a, b = BatchSim(model_it1, n, lib_path=emul_lib), BatchSim(model_it1, n, lib_path=emul_lib)
    seeds = (20 + np.arange(n)).astype(np.uint64)
    a.reset(seeds + np.uint64(100), 1, 40.0)                      # the controller's PID state persists across resets: same history
    b.reset(seeds + np.uint64(100), 1, 40.0)
    b.move_group(1 << 6, [0.1], 0.05, 30)                         # so that last_movement_steps is not 0 before the reset
    a.move_group(1 << 6, [0.1], 0.05, 30)
    a.reset(seeds, 1, 40.0)
    before, cb = b.get_state(), b.counters()
    assert cb["last_steps"][1] > 0
    mask = np.array([1, 0, 1, 1], dtype=np.uint8)
    b.reset_dev(seeds.ctypes.data, mask.ctypes.data, 40.0)        # emulation build: "device" pointers are host pointers
    b.sync()
    sa, sb = a.get_state(), b.get_state()
    for k in ("qpos", "qvel", "warmstart"):
        assert np.array_equal(sa[k][mask == 1], sb[k][mask == 1]), k
        assert np.array_equal(before[k][1], sb[k][1]), k
    ca = b.counters()                                             # an unflagged scene keeps its whole record, counters included
    assert ca["last_steps"][1] == cb["last_steps"][1] and ca["total_steps"][1] == cb["total_steps"][1]


def test_host_step_skip_uses_the_kernel_early_out(model_it1, emul_lib):
    """GraspingEnv.py:124-131: a skipped scene keeps its whole record (ctrl and counters included), reward 0."""
    import numpy as np
    from mujoco_rl_ur5_amd.native import BatchSim
    sim = BatchSim(model_it1, 2, lib_path=emul_lib)
    sim.reset((20 + np.arange(2)).astype(np.uint64), 1, 40.0)
    st0, c0, u0 = sim.get_state(), sim.counters(), sim.get_ctrl()
    rew, ps, _ = sim.grasp_attempt(np.array([[0.0, -0.6, 0.95], [0.0, -0.6, 0.95]]), rot=0, check_mode=1, skip=np.array([1, 0]))
    st1, c1, u1 = sim.get_state(), sim.counters(), sim.get_ctrl()
    assert rew[0] == 0 and ps[0].sum() == 0 and c1["total_steps"][0] == c0["total_steps"][0] and np.array_equal(u0[0], u1[0])
    assert all(np.array_equal(st0[k][0], st1[k][0]) for k in st0)
    assert c1["total_steps"][1] > c0["total_steps"][1]


def test_attempt_plus_reset_in_one_launch_equals_two_calls(model_it1, emul_lib):
    """ur5_grasp_attempt_reset_dev == ur5_grasp_attempt_dev followed by ur5_reset_dev for the scenes with a non-zero seed, bit for bit;
    a dispatch order (ur5_set_order_dev) changes nothing."""
    import numpy as np
    from mujoco_rl_ur5_amd.native import BatchSim
    n = 3
    seeds0 = (20 + np.arange(n)).astype(np.uint64)
    act = np.zeros((n, 8))
    act[:, :3] = [0.0, -0.6, 0.95]
    new_seeds = np.array([77, 0, 79], dtype=np.uint64)
    order = np.array([2, 0, 1], dtype=np.int32)
    out = []
    for fused in (True, False):
        sim = BatchSim(model_it1, n, lib_path=emul_lib)
        sim.reset(seeds0, 1, 60.0)
        rew = np.zeros(n, dtype=np.int32)
        st = sim.get_state()
        st["qvel"][0, 2] = np.nan                                 # scene 0 blows up in its first step: flagged (status bit 2), then reset by its new seed
        st["qvel"][1, 2] = np.nan                                 # scene 1 too, but it is not reset: its flag must survive the launch
        sim.set_state(qvel=st["qvel"])
        if fused:
            tmp = order.copy()
            sim.set_order_dev(tmp.ctypes.data)
            tmp[:] = -7                                           # the handle keeps its own copy: the caller's buffer may die after the call
            del tmp
            sim.grasp_attempt_reset_dev(act.ctypes.data, rew.ctypes.data, new_seeds.ctypes.data, check_mode=1, settle_ms=60.0)
        else:
            sim.grasp_attempt_dev(act.ctypes.data, rew.ctypes.data, check_mode=1)
            mask = (new_seeds != 0).astype(np.uint8)
            sim.reset_dev(new_seeds.ctypes.data, mask.ctypes.data, 60.0)
        sim.sync()
        out.append((rew.copy(), sim.get_state(), sim.counters()["total_steps"].copy(), sim.counters()["status"].copy(), sim.counters()["status_ended"].copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])
    # status bits are sticky until a reset -- the fused reset (script state 19) clears them exactly like ur5_reset_dev
    assert out[0][3].tolist() == out[1][3].tolist() == [0, 2, 0]
    # ... but the flag of the attempt whose reward the fused launch returned stays readable (bits 8-15 of the status column) until a host-side reset;
    # with two calls the caller can read the counters in between, and ur5_reset_dev -- an explicit host action -- clears everything
    assert out[0][4].tolist() == [2, 0, 0] and out[1][4].tolist() == [0, 0, 0]
    for k in out[0][1]:
        assert np.array_equal(out[0][1][k], out[1][1][k]), k


def test_demo_mode_runs_the_100_step_final_check(model_it1, emul_lib):
    """GraspingEnv.py:313-321: with demo=True the closing check at the drop position is close_gripper(max_steps=100), not 1000 -- a `max steps`
    result after 100 steps already counts as "object in the gripper". GraspEnv(demo=True) selects check_mode 2 of the engine; the engine's
    script and the oracle's agree on it, and the check phase really lasts 100 steps when something was grasped."""
    import numpy as np
    from conftest import aimed_actions
    from mujoco_rl_ur5_amd.envs import GraspEnv
    from oracle.oracle import Oracle
    env = GraspEnv(file=model_it1, n_envs=1, show_obs=False, observation="flat", demo=True, _lib_path=emul_lib)
    assert env.check_mode == 2 and GraspEnv(file=model_it1, n_envs=1, show_obs=False, observation="flat", _lib_path=emul_lib).check_mode == 0
    sim = env.sim
    sim.reset([20], 1, 1000.0)
    acts = aimed_actions(sim.get_state()["qpos"], 4)
    rew, ps, pr = sim.grasp_attempt(acts, rot=0, check_mode=env.check_mode)
    o = Oracle(model_it1)
    o.reset(20, 1, True)
    r, pso, pro = o.grasp_attempt(acts[0], 0, 2)
    assert r == rew[0] and pso.tolist() == ps[0].tolist() and pro.tolist() == pr[0].tolist()
    assert pr[0][5] == 1 and ps[0][9] == 101 and pr[0][9] == 1 and rew[0] == 1      # closed on the box (phase 5 timed out); final check: max_steps = 100 (the loop reports max_steps + 1, MujocoController.py:351-364), still closed
    o = Oracle(model_it1)                                                           # (a fresh controller: the PID state persists across resets)
    o.reset(20, 1, True)
    r0, ps0, _ = o.grasp_attempt(acts[0], 0, 0)
    assert ps0[9] == 1001 and ps0[5] == 301 and ps0[:9].tolist() == pso[:9].tolist()                   # the non-demo script differs only in that phase (and what follows it)
