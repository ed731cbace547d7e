"""ur5_grasp_rounds_dev: K consecutive rounds of every scene in ONE launch, the scripted aiming rule evaluated in the kernel, no lock step between scenes.

The reference's episode loop (example_agent.py:15-27) has no barrier between scenes -- it has one scene. For a scripted policy that reads only the scene's own state
the engine may run a scene's rounds back to back; what must hold is that every per-scene result is the one the lock-step shape (one launch per round, the action
computed outside from the state the previous launch left: bench.It1Rounds.launch) produces: every word of the state records, every reward, every action record."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mujoco_rl_ur5_amd.native import BatchSim  # noqa: E402


def _both_ways(model, n, rounds, fused, dev, lo=0, n_total=None, kind="it1", **kw):
    """`rounds` rounds of scenes lo .. lo + n - 1, once as one launch per round and once as launches of `fused` rounds: (records, rewards, actions, counters,
    observations) of both. kind "it4" / "many": the rendered workloads -- lock step = stand-alone render, torch rule, depth lookup, launch; fused = every scene renders
    for itself and aims inside the launch (ur5_set_observation_dev, rule z_from_depth / kind 2)."""
    n_total = n_total or n
    out = []
    for k in (0, fused):
        sim = BatchSim(model, n, **kw)
        sim.reset((bench.BASE_SEED + lo + np.arange(n)).astype(np.uint64), 1, 1000.0)
        if dev.type == "cuda":
            sim.set_stream(torch.cuda.current_stream().cuda_stream)
        wl = bench.It1Rounds(torch, model, sim, dev, lo, n, n_total, "aimed", kind)
        rew = torch.zeros((rounds, n), dtype=torch.int32, device=dev)
        acts, pixels, frames = [], [], []
        r = 0
        while r < rounds:
            if k == 0:
                a, px = wl.launch(r, rew[r])
                sim.sync()
                acts.append(a[None, :, :4].clone()); pixels.append(px[None].clone())
                if kind != "it1":
                    frames.append((wl.img[None].clone(), wl.dep[None].clone()))          # the observation round r was chosen in
                r += 1
            else:
                kk = min(k, rounds - r)
                a, px = wl.launch_rounds(r, kk, rew[r:r + kk])
                sim.sync()
                acts.append(a[:, :, :4].clone()); pixels.append(px.clone())
                if kind != "it1":
                    frames.append((wl._frames[0][:kk].clone(), wl._frames[1][:kk].clone()))
                r += kk
        c = sim.counters()
        obs = (torch.cat([f[0] for f in frames]).cpu().numpy(), torch.cat([f[1] for f in frames]).cpu().numpy()) if frames else None
        out.append((wl.state.clone().cpu().numpy(), rew.cpu().numpy(), torch.cat(acts).cpu().numpy(), torch.cat(pixels).cpu().numpy(),
                    {key: c[key].copy() for key in ("total_steps", "solver_iters", "status", "status_ended", "ncon_max", "last_steps")}, obs))
        sim.close()
    return out


def _assert_identical(a, b, rounds, n):
    assert a[1].shape == (rounds, n) and np.array_equal(a[1], b[1]), "rewards"
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), "action records / aimed pixels"
    assert np.array_equal(a[0], b[0]), "state records: every word (qpos, qvel, warm start, controls, PID state, counters)"
    for key in a[4]:
        assert np.array_equal(a[4][key], b[4][key]), key


def test_fused_rounds_equal_lock_step_rounds_on_the_emulation_build(model_it1, emul_lib):
    """Lane emulation of the engine source, 4 scenes x 6 rounds (every scene crosses an episode boundary, two of them two), launches of 4 + 2 rounds against 6 launches."""
    n, rounds = 4, 6
    a, b = _both_ways(model_it1, n, rounds, 4, torch.device("cpu"), lib_path=emul_lib)
    _assert_identical(a, b, rounds, n)
    assert a[1].mean() > 0.4 and a[4]["total_steps"].min() > 6 * 1200            # attempts that really grasp, resets that really settle
    # a shard of a larger job: global scene ids and the job's scene count enter the rule and the episode seeds
    a, b = _both_ways(model_it1, 2, 4, 4, torch.device("cpu"), lo=6, n_total=16, lib_path=emul_lib)
    _assert_identical(a, b, 4, 2)


def test_fused_rounds_on_the_device_code_path(model_it1, simt_lib):
    """The interpreter as the GPU runs it (wavefront intrinsics, one fibre per lane) on the SIMT host build: two rounds of one scene in one launch against two launches
    (the episode boundary inside a launch is the emulation build's and the GPU test's case: both run this same interpreter)."""
    a, b = _both_ways(model_it1, 1, 2, 2, torch.device("cpu"), lo=0, n_total=4, lib_path=simt_lib)
    _assert_identical(a, b, 2, 1)
    assert a[4]["total_steps"].min() > 2 * 1200


def _assert_same_observations(a, b):
    """the frames a fused launch leaves behind are the stand-alone renders of the lock-step rounds: same ray caster on the same states"""
    assert a[5][0].shape == b[5][0].shape and np.array_equal(a[5][0], b[5][0]), "RGB frames"
    assert np.array_equal(a[5][1].view(np.uint32), b[5][1].view(np.uint32)), "depth frames"
    assert a[5][1].min() > 0.5 and a[5][1].max() > 1.0 and len(np.unique(a[5][0].reshape(-1, 3), axis=0)) > 10       # real images


def test_rendered_rounds_inside_the_launch_on_the_emulation_build(model_2f, emul_lib):
    """Round 6, BASELINE configs[2] shape (six-object scene, 200x200 RGB-D observation per round, grasp height from the depth under the aimed pixel): 2 scenes x 5 rounds,
    launches of 3 + 2 rounds with the observation rendered and the rule evaluated INSIDE the launch against five lock-step rounds of ur5_render_dev + torch + one launch."""
    n, rounds = 2, 5
    a, b = _both_ways(model_2f, n, rounds, 3, torch.device("cpu"), kind="it4", lib_path=emul_lib)
    _assert_identical(a, b, rounds, n)
    _assert_same_observations(a, b)
    assert a[2][..., 2].min() > 0.9 and np.ptp(a[2][..., 2]) > 1e-3                 # grasp heights read from the depth images, not the constant 0.91


def test_pile_rounds_inside_the_launch_on_the_emulation_build(emul_lib):
    """... and configs[3]: 40-object piles, the box rule (kind 2) evaluated by the scene, 2 scenes x 2 rounds in one launch against two lock-step rounds."""
    from mujoco_rl_ur5_amd.model import load_model
    m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
    a, b = _both_ways(m, 2, 2, 2, torch.device("cpu"), kind="many", lib_path=emul_lib)
    _assert_identical(a, b, 2, 2)
    _assert_same_observations(a, b)


def test_rules_go_with_their_scenes(model_it1, emul_lib):
    from mujoco_rl_ur5_amd.model import load_model
    from mujoco_rl_ur5_amd.native import AimRule
    rew = np.zeros(2, dtype=np.int32)
    pile = BatchSim(load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml"), 1, lib_path=emul_lib)
    with pytest.raises(RuntimeError, match="rule kind 2"):                          # the plate rule reads slide-joint boxes on the pick plate: not a pile's
        pile.grasp_rounds_dev(AimRule(kind=1, episode_rounds=4, first_scene_id=0, n_total=1, base_seed=20), 0, 2, rew.ctypes.data)
    with pytest.raises(RuntimeError, match="ur5_set_observation_dev"):               # a height from the depth image needs the image
        pile.grasp_rounds_dev(AimRule(kind=2, episode_rounds=4, first_scene_id=0, n_total=1, base_seed=20, z_from_depth=1, cam_dx=1.0, cam_dy=1.0), 0, 2, rew.ctypes.data)
    small = BatchSim(model_it1, 1, lib_path=emul_lib)
    with pytest.raises(RuntimeError, match="40-object"):
        small.grasp_rounds_dev(AimRule(kind=2, episode_rounds=4, first_scene_id=0, n_total=1, base_seed=20), 0, 2, rew.ctypes.data)
    buf = np.zeros(200 * 200 * 4, dtype=np.uint8)
    with pytest.raises(RuntimeError, match="no room"):                              # the four-box image has no room for the ray caster's working set: refused, not skipped
        small.set_observation_dev(buf.ctypes.data, buf.ctypes.data, 0, 200, 200)


@pytest.mark.gpu
def test_fused_rounds_equal_lock_step_rounds_on_gpu(model_it1):
    """Round-4 verdict item 4: 64 scenes x 8 rounds both ways on the MI355X, every record word equal."""
    n, rounds = 64, 8
    a, b = _both_ways(model_it1, n, rounds, 4, torch.device("cuda", 0), device_id=0)
    _assert_identical(a, b, rounds, n)
    assert 0.45 < a[1].mean() < 0.9 and a[4]["status"].max() == 0


@pytest.mark.gpu
def test_rendered_rounds_inside_the_launch_on_gpu(model_2f):
    """The rendered workloads on the MI355X: 32 six-object scenes x 6 rounds as launches of 3 rounds (observation + rule inside the launch) against six lock-step rounds
    (ur5_render_dev, torch rule, depth lookup, one launch each): every record word, reward, action record and aimed pixel equal; the frames a launch leaves behind are the
    stand-alone renders (RGB within one count, depth to 1e-6 m: the two kernels compile the same ray caster separately)."""
    n, rounds = 32, 6
    a, b = _both_ways(model_2f, n, rounds, 3, torch.device("cuda", 0), kind="it4", device_id=0)
    _assert_identical(a, b, rounds, n)
    assert np.abs(a[5][0].astype(np.int32) - b[5][0].astype(np.int32)).max() <= 1 and np.abs(a[5][1] - b[5][1]).max() < 1e-6
    assert a[4]["status"].max() == 0 and 0.2 < a[1].mean() < 0.9


@pytest.mark.gpu
def test_pile_rounds_inside_the_launch_on_gpu():
    """... and 16 piles x 3 rounds in one launch: the box rule evaluated by the scene itself (rule kind 2) from the depth image it rendered."""
    from mujoco_rl_ur5_amd.model import load_model
    m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
    a, b = _both_ways(m, 16, 3, 3, torch.device("cuda", 0), kind="many", device_id=0)
    _assert_identical(a, b, 3, 16)
    assert np.abs(a[5][0].astype(np.int32) - b[5][0].astype(np.int32)).max() <= 1 and np.abs(a[5][1] - b[5][1]).max() < 1e-6
    assert a[4]["status"].max() == 0
