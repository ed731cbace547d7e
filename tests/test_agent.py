"""The batched DQN loop (mujoco_rl_ur5_amd/agent.py): CPU run on the lane-emulation engine with host
tensors; the same code on cuda:0 with -m gpu."""
import numpy as np
import pytest
import torch

from mujoco_rl_ur5_amd.agent import BATCH_SIZE, BatchedGraspAgent
from mujoco_rl_ur5_amd.envs import GraspEnv


def _checks(agent, out):
    N = agent.N
    assert out["reward"].shape == (N,) and set(out["reward"].tolist()) <= {0, 1}
    assert out["outcomes"].shape == (N, 4) and out["outcomes"].dtype == torch.int32
    assert (out["outcomes"][:, 1] >= 0).all() and (out["outcomes"][:, 1] < agent.n_actions_1).all() and (out["outcomes"][:, 2] < 6).all()


def test_device_observation_equals_host_observation(model_it1, emul_lib):
    env = GraspEnv(file=model_it1, n_envs=2, show_obs=False, observation="render", image_width=40, image_height=40, _lib_path=emul_lib)
    host = env.reset()
    dev = env.observation_device("cpu")
    assert np.array_equal(dev["rgb"].numpy(), host["rgb"]) and np.allclose(dev["depth"].numpy(), host["depth"], rtol=0, atol=1e-6)
    # pixel_2_world for every pixel at once == the controller's scalar version (MujocoController.py:783-806)
    w = env.pixel_world_device(dev["depth"], "cpu")
    ref = env.controller.pixel_2_world(pixel_x=7, pixel_y=31, depth=float(host["depth"][1, 31, 7]), height=40, width=40)
    assert np.allclose(w[1, 31, 7].numpy(), ref, atol=1e-9)
    # step_device == step on the same action
    a = np.array([[20 * 40 + 20, 1], [5 * 40 + 33, 4]])
    env2 = GraspEnv(file=model_it1, n_envs=2, show_obs=False, observation="render", image_width=40, image_height=40, _lib_path=emul_lib)
    env2.reset()
    _, r_host, _, info = env2.step(a)
    r_dev, skipped = env.step_device(torch.from_numpy(a), dev["depth"], "cpu")
    assert np.array_equal(r_dev.numpy(), r_host) and np.array_equal(skipped.numpy(), info["skipped"])
    assert np.allclose(env.sim.get_state()["qpos"], env2.sim.get_state()["qpos"], atol=1e-12)


def test_agent_rounds_on_cpu(model_it1, emul_lib):
    env = GraspEnv(file=model_it1, n_envs=2, show_obs=False, observation="render", image_width=24, image_height=24, check_mode=1, _lib_path=emul_lib)
    env.reset()
    agent = BatchedGraspAgent(env=env, device="cpu", mem_size=64)
    obs = env.observation_device("cpu")
    state = agent.transform_observation(obs, jitter_and_noise=False)
    assert state.shape == (2, 4, 24, 24) and float(state.min()) >= 0 and float(state.max()) <= 1
    d = obs["depth"].clamp(max=agent.depth_threshold)                       # transform_observation of the reference, one image
    ref = (-d[0] - (-d[0]).min()) / ((-d[0]).max() - (-d[0]).min())
    # the reference adds N(0, 1e-3) depth noise whenever normalize=True (Grasping_Agent_multidiscrete.py:317): 5 sigma over the ~0.1-0.2 m range
    span = float((-d[0]).max() - (-d[0]).min())                           # a view of the bare table has no depth range: the noise is all there is
    assert (span < 0.02 or torch.allclose(state[0, 3], ref, atol=0.006 / span)) and torch.allclose(state[0, :3], obs["rgb"][0].permute(2, 0, 1).float() / 255)
    action, greedy = agent.epsilon_greedy(state, obs)                       # epsilon = 1 at the start: every action is a random table pixel
    assert not greedy.any() and agent.steps_done == 2
    pa = agent.transform_action(action)
    world = env.pixel_world_device(obs["depth"], "cpu")
    for e in range(2):
        assert world[e, pa[e, 0] // 24, pa[e, 0] % 24, 2] >= env.TABLE_HEIGHT - 0.01
    agent.eps_start = agent.eps_end = 0.0                                   # greedy: argmax of the Q maps
    action, greedy = agent.epsilon_greedy(state, obs)
    with torch.no_grad():
        q1 = agent.policy_net(state[1:2])                                   # the reference's select_action: ONE observation, network in training mode (:232-299)
    assert greedy.all() and int(action[1]) == int(q1.reshape(-1).argmax())
    # ... and a scene's Q map does not depend on which scenes share its forward pass (batch norm with per-image statistics, qnet.per_sample_statistics)
    v2, i2 = agent._q_all(state)
    v1, i1 = agent._q_all(state, chunk=1)
    assert torch.equal(i1, i2) and torch.allclose(v1, v2, atol=1e-6) and abs(float(v2[1]) - float(q1.max())) < 1e-6
    agent.eps_start = agent.eps_end = 1.0
    losses = []
    for r in range(2 * BATCH_SIZE // 2 + 1):
        out = agent.round()
        _checks(agent, out)
        losses.append(out["loss"])
    assert losses[0] is None and losses[-1] is not None and np.isfinite(losses[-1]) and len(agent.memory) == 26


@pytest.mark.gpu
def test_agent_rounds_on_gpu(model_it1):
    assert torch.cuda.is_available()
    env = GraspEnv(file=model_it1, n_envs=8, show_obs=False, observation="render", check_mode=1)
    env.reset()
    agent = BatchedGraspAgent(env=env, device="cuda", mem_size=64)
    obs = env.observation_device("cuda")
    assert obs["rgb"].is_cuda and obs["depth"].is_cuda and obs["depth"].shape == (8, 200, 200)
    host = env.get_observation(show=False)
    assert np.array_equal(obs["rgb"].cpu().numpy(), host["rgb"]) and np.allclose(obs["depth"].cpu().numpy(), host["depth"], atol=1e-6)
    for r in range(4):
        out = agent.round()
        _checks(agent, out)
        assert out["reward"].is_cuda
    assert out["loss"] is not None and np.isfinite(out["loss"]) and len(agent.memory) == 32
    assert env.sim.counters()["status"].max() == 0


def test_color_jitter_is_torchvisions_colorjitter_for_a_batch():
    """ColorJitter(0.5, 0.5, 0.5, 0.5) of Grasping_Agent_multidiscrete.py:120-126 on the device: torchvision's tensor definitions of the four
    operations, per-image factors and operation order."""
    from mujoco_rl_ur5_amd.agent import color_jitter, _rgb_to_hsv, _hsv_to_rgb, _gray
    g = torch.Generator().manual_seed(5)
    x = torch.rand(6, 3, 8, 8, generator=g)
    assert torch.allclose(_hsv_to_rgb(_rgb_to_hsv(x)), x, atol=1e-6)                      # HSV round trip
    red = torch.zeros(1, 3, 2, 2); red[:, 0] = 1.0
    hsv = _rgb_to_hsv(red); hsv[:, 0] = (hsv[:, 0] + 1.0 / 3.0) % 1.0
    assert torch.allclose(_hsv_to_rgb(hsv), torch.tensor([0.0, 1.0, 0.0]).view(1, 3, 1, 1).expand(1, 3, 2, 2), atol=1e-6)   # red + 120 deg = green
    assert torch.allclose(color_jitter(x, torch.Generator().manual_seed(1), 0.0, 0.0, 0.0, 0.0), x, atol=1e-6)   # all factors 1 / 0: identity
    y = color_jitter(x, torch.Generator().manual_seed(1))
    assert y.shape == x.shape and float(y.min()) >= 0 and float(y.max()) <= 1 and not torch.allclose(y, x)
    # brightness only: every image is its input times one factor in [0.5, 1.5] (clamped), a different one per image
    yb = color_jitter(0.5 * x, torch.Generator().manual_seed(2), 0.5, 0.0, 0.0, 0.0)
    f = (yb.flatten(1).sum(1) / (0.5 * x).flatten(1).sum(1))
    assert torch.allclose(yb, (0.5 * x) * f.view(6, 1, 1, 1), atol=1e-5) and float(f.min()) >= 0.5 - 1e-5 and float(f.max()) <= 1.5 + 1e-5
    assert f.unique().numel() == 6
    # saturation 0 end of the range collapses to the grey image (torchvision: blend(img, gray, s))
    assert torch.allclose(0.0 * x + 1.0 * _gray(x).expand_as(x), _gray(x).expand_as(x))


@pytest.mark.gpu
def test_graspenv_step_on_gpu_host_path_equals_device_path_and_the_oracle(model_it1):
    """GraspEnv.step (GraspingEnv.py:62-156) on the MI355X through its HOST surface -- pixel action, depth lookup, back-projection, skip rule,
    one kernel launch, reward, new observation -- against (a) step_device on the same actions and (b) the oracle's move_and_grasp from the
    same coordinates for the non-skipped scenes."""
    from oracle.oracle import Oracle
    n = 6
    a = GraspEnv(file=model_it1, n_envs=n, show_obs=False, observation="render")
    b = GraspEnv(file=model_it1, n_envs=n, show_obs=False, observation="render")
    a.reset(); b.reset()
    q = a.sim.get_state()["qpos"]
    act = np.zeros((n, 2), dtype=np.int64)
    for e in range(n):
        k = e % 4
        px, py = a.controller.world_2_pixel([q[e][8 + 7 * k], -0.6 + q[e][8 + 7 * k + 1], 0.951])
        act[e] = [py * 200 + px, e % 6]
    act[5] = [3 * 200 + 3, 0]                                                 # a corner pixel: the floor beside the table -> skipped (:124)
    obs, reward, done, info = a.step(act)
    dev = b.observation_device()
    r_dev, skipped = b.step_device(torch.from_numpy(act).cuda(), dev["depth"])
    assert done is False and obs["depth"].shape == (n, 200, 200) and np.array_equal(reward, r_dev.cpu().numpy())
    assert info["skipped"].tolist() == skipped.cpu().tolist() == [False] * 5 + [True]
    # the two paths back-project the pixel in different arithmetic (numpy on the host, torch on the device): last-bit differences of the target
    assert np.allclose(a.sim.get_state()["qpos"][:, :8], b.sim.get_state()["qpos"][:, :8], atol=1e-6)
    x, y = act[:, 0] % 200, act[:, 0] // 200
    depth0 = dev["depth"].cpu().numpy()
    coords = a.controller.pixel_2_world_batch(x, y, depth0[np.arange(n), y, x])
    for e in (0, 3):
        o = Oracle(model_it1)
        o.reset(20 + e, 1, True)
        r, ps, pr = o.grasp_attempt(coords[e], int(act[e, 1]), 0)
        assert r == reward[e] and ps.tolist() == info["phase_steps"][e].tolist()


def _loop(model, lib, groups, rounds=6, n=6, device="cpu", width=24, **kw):
    """`rounds` rounds of the config-5 loop with `groups` pipelined scene groups: (outcome records, losses, weight digest, ring bookkeeping, optimiser steps)."""
    import hashlib
    torch.manual_seed(0)
    common = dict(file=model, show_obs=False, observation="render", image_width=width, image_height=width, check_mode=1, **({"_lib_path": lib} if lib else {}))
    if groups == 1:
        env = GraspEnv(n_envs=n, **common)
        agent = BatchedGraspAgent(env=env, device=device, mem_size=40, eps_start=0.5, eps_end=0.5, max_updates_per_round=4, **kw)
    else:
        agent = BatchedGraspAgent(n_envs=n, device=device, mem_size=40, eps_start=0.5, eps_end=0.5, max_updates_per_round=4, pipeline_groups=groups, **common, **kw)
    for e in agent.envs:
        e.reset()
    recs, losses = [], []
    for _ in range(rounds):
        out = agent.round()
        recs.append(out["outcomes"].cpu().numpy().copy())
        losses += out["losses"]
    w = torch.cat([p.detach().reshape(-1).cpu() for p in agent.policy_net.parameters()] + [b.detach().reshape(-1).float().cpu() for b in agent.policy_net.buffers()])
    states = np.concatenate([e.sim.get_state()["qpos"] for e in agent.envs])
    return np.stack(recs), losses, hashlib.sha256(w.numpy().tobytes()).hexdigest(), (agent.memory.position, agent.memory.count), agent.learner.updates_done, states


def test_pipelined_scene_groups_are_the_same_agent(model_it1, emul_lib):
    """Round-4 verdict item 2: the config-5 loop with two (three) scene groups per rank -- group g + 1's render, CNN forward and action selection queued under group g's
    grasp launch, a group's replay pushes and optimiser steps under the next group's launch -- must BE the unpipelined loop: the same outcome records in every round,
    the same loss sequence (chunks of the round's chunking that straddle two groups included: 6 scenes, 4 updates per round -> chunks of 2; 3 groups of 2, 2 groups
    of 3), bit-identical weights, the same replay ring, the same final scene states."""
    one = _loop(model_it1, emul_lib, 1)
    for g in (2, 3):
        many = _loop(model_it1, emul_lib, g)
        assert np.array_equal(one[0], many[0]) and one[0][:, :, 0].tolist() == [list(range(6))] * 6
        assert one[1] == many[1] and len(one[1]) == one[4] == many[4] >= 6
        assert one[2] == many[2] and one[3] == many[3] and np.array_equal(one[5], many[5])


@pytest.mark.gpu
def test_pipelined_scene_groups_on_gpu(model_it1):
    """The same on the MI355X with real streams (one engine handle + CUDA stream per group, 64 x 64 observations): outcome records of the rounds before the first
    optimiser step equal the unpipelined loop's by construction; afterwards the two runs differ only by MIOpen's run-to-run rounding (first losses to 1e-4 / 5 %)."""
    one = _loop(model_it1, None, 1, rounds=5, n=8, device="cuda", width=64)
    two = _loop(model_it1, None, 2, rounds=5, n=8, device="cuda", width=64)
    assert np.array_equal(one[0][:3], two[0][:3]) and one[4] == two[4] >= 6 and one[3] == two[3]
    assert abs(one[1][0] - two[1][0]) < 1e-4 * one[1][0] and np.allclose(one[1][:2], two[1][:2], rtol=0.05)


@pytest.mark.gpu
def test_late_group_selects_with_the_round_start_weights(model_it1):
    """Round-5 advisor finding: group g's CNN forward runs on its own stream; nothing but an explicit event keeps the learner's optimiser steps (main stream, started as soon as
    group 0's launch has run) from overwriting policy_net under a LATE group's forward. The hook delays groups 1.. by ~50 ms of GPU time and reads a flag that the main stream
    sets right in front of its first optimiser step: every group must have chosen its actions before that flag is set (stream order, not numerics)."""
    torch.manual_seed(0)
    common = dict(file=model_it1, show_obs=False, observation="render", image_width=64, image_height=64, check_mode=1)
    agent = BatchedGraspAgent(n_envs=8, device="cuda", mem_size=40, eps_start=0.5, eps_end=0.5, max_updates_per_round=4, pipeline_groups=2, **common)
    for e in agent.envs:
        e.reset()
    agent._order_probe = dict(flag=torch.zeros(1, dtype=torch.int32, device="cuda"), delay_cycles=100_000_000, seen=[])
    for _ in range(5):                                                                            # (the ring holds 2 x batch transitions after three rounds of 8: the learner steps from then on)
        out = agent.round()
    torch.cuda.synchronize()
    seen = torch.cat(agent._order_probe["seen"]).cpu().tolist()
    assert len(seen) == 10 and seen == [0] * 10, seen
    assert agent.learner.updates_done >= 4 and int(agent._order_probe["flag"].item()) == 1        # the learner did run, behind the flag


def test_checkpoint_is_the_reference_trainers(model_it1, emul_lib, tmp_path):
    """save() writes the dict of Grasping_Agent_multidiscrete.py:560-575 (same keys; rotation counters as str -> int dicts, :460-465) and ``load_path`` resumes from it the way
    the reference's constructor does (:109-114, :157-180): weights, optimiser moments, step count, epsilon, counters."""
    common = dict(file=model_it1, show_obs=False, observation="render", image_width=24, image_height=24, check_mode=1, _lib_path=emul_lib)
    torch.manual_seed(0)
    agent = BatchedGraspAgent(env=GraspEnv(n_envs=6, **common), device="cpu", mem_size=40, eps_start=0.5, eps_end=0.5, max_updates_per_round=4)
    agent.env.reset()
    for _ in range(5):
        out = agent.round()
    assert agent.learner.updates_done >= 4
    path = str(tmp_path / "weights.pt")
    agent.save(path)
    ck = torch.load(path)
    assert sorted(ck) == sorted(["step", "model_state_dict", "optimizer_state_dict", "epsilon", "greedy_rotations", "greedy_rotations_successes", "random_rotations_successes"])
    assert ck["step"] == agent.steps_done == 30 and ck["epsilon"] == agent.eps_threshold
    assert sum(ck["greedy_rotations"].values()) == int(sum(int(o) for o in [agent._rot_counts[0].sum()])) and all(isinstance(k, str) for k in ck["greedy_rotations"])
    twin = BatchedGraspAgent(env=GraspEnv(n_envs=6, **common), device="cpu", mem_size=40, eps_start=0.5, eps_end=0.5, max_updates_per_round=4, load_path=path)
    for a, b in zip(agent.policy_net.state_dict().values(), twin.policy_net.state_dict().values()):
        assert torch.equal(a, b)
    sa, sb = agent.optimizer.state_dict()["state"], twin.optimizer.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) for k in sa)
    assert twin.steps_done == 30 and twin.eps_threshold == agent.eps_threshold and torch.equal(twin._rot_counts, agent._rot_counts)
