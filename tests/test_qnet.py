"""The grasp-Q CNN and replay buffer against vectors produced by the reference's own Modules.py
(tools/gen_golden_qnet.py -> tests/golden/qnet_reference.json)."""
import json
import os

import numpy as np
import pytest
import torch

from mujoco_rl_ur5_amd import qnet

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "qnet_reference.json")))


@pytest.mark.parametrize("tag,make", [("MULTIDISCRETE_RESNET_6", lambda: qnet.MULTIDISCRETE_RESNET(6)), ("RESNET", qnet.RESNET),
                                      ("POLICY_RESNET", qnet.POLICY_RESNET)])
def test_network_equals_the_reference(tag, make):
    g = GOLD[tag]
    torch.manual_seed(0)
    net = make().eval()
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == g["keys"]      # the reference's checkpoints load unchanged
    assert qnet.count_parameters(net) == g["n_params"]
    x = torch.randn(2, 4, 40, 40, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = net(x)
    assert list(y.shape) == g["out_shape"]
    flat = y.reshape(-1)
    assert np.allclose(flat[torch.tensor(g["sample_idx"])].numpy(), g["sample"], rtol=1e-5, atol=1e-6)
    assert abs(float(flat.double().sum()) - g["sum"]) < 1e-3 * max(1.0, abs(g["sum"]))
    assert abs(float(flat.double().abs().sum()) - g["abs_sum"]) < 1e-3 * g["abs_sum"]


def test_train_mode_forward_equals_the_reference():
    torch.manual_seed(0)
    net = qnet.MULTIDISCRETE_RESNET(6).train()
    y = net(torch.randn(3, 4, 40, 40, generator=torch.Generator().manual_seed(2)))
    assert abs(float(y.double().sum()) - GOLD["train_mode"]["sum"]) < 1e-3 * abs(GOLD["train_mode"]["sum"])


def test_single_image_loses_its_batch_axis_like_the_reference():
    net = qnet.MULTIDISCRETE_RESNET(6).eval()
    with torch.no_grad():
        assert net(torch.zeros(1, 4, 40, 40)).shape == (6, 40, 40)            # x.squeeze_() in Modules.py:277


def test_replay_buffer_semantics():
    g = GOLD["replay"]
    buf = qnet.ReplayBuffer(5, height=4, width=4)
    for i in range(8):                                                           # the reference pushed states 0..7 into a buffer of 5
        s = torch.zeros(1, 4, 4, 4)
        s[0, 0, 0, 0] = i / 255.0
        buf.push(s, torch.tensor([10 * i]), torch.tensor([i % 2]))
    assert len(buf) == g["len"] and buf.position == g["position"]
    assert [int(round(float(v) * 255)) for v in (buf.rgb[:, 0, 0, 0].float() / 255.0)] == g["stored_states"]
    for _ in range(4):
        state, action, reward = buf.sample(3)
        assert state.shape == (3, 4, 4, 4) and action.shape == (3, 1) and reward.shape == (3, 1)
        assert int(round(float(state[-1, 0, 0, 0]) * 255)) == g["sample_last"][0]   # the newest transition is always in the batch
        assert int(action[-1]) == 10 * g["sample_last"][0]
    nb = qnet.ReplayBuffer(6, height=4, width=4)                                  # batched push = consecutive pushes in scene order
    nb.push(torch.zeros(4, 4, 4, 4), torch.arange(4), torch.zeros(4))
    nb.push(torch.zeros(4, 4, 4, 4), 4 + torch.arange(4), torch.ones(4))
    assert nb.position == 2 and len(nb) == 6 and nb.action[:, 0].tolist() == [6, 7, 2, 3, 4, 5]


def _learn_step(device):
    """Grasp_Agent.learn() (Grasping_Agent_multidiscrete.py:388-446) with this package's network and BatchedGraspAgent's loss / optimiser
    settings on the batch the reference's Modules.py was run on (tools/gen_golden_qnet.py "learn_step")."""
    import torch.nn.functional as F
    g = GOLD["learn_step"]
    torch.manual_seed(0)
    net = qnet.MULTIDISCRETE_RESNET(6).train()
    gen = torch.Generator().manual_seed(3)
    state = torch.rand(12, 4, 40, 40, generator=gen)
    action = torch.randint(0, 6 * 40 * 40, (12, 1), generator=gen)
    reward = torch.randint(0, 2, (12, 1), generator=gen)
    net, state, action, reward = net.to(device), state.to(device), action.to(device), reward.to(device)
    opt = torch.optim.Adam(net.parameters(), lr=0.001, weight_decay=0.00002)
    q_pred = net(state).reshape(12, -1).gather(1, action)
    loss = F.binary_cross_entropy(q_pred, reward.float())
    loss.backward()
    opt.step()
    opt.zero_grad()
    net.eval()
    with torch.no_grad():
        y2 = net(state[:2])
    return g, float(loss), q_pred.detach().reshape(-1).cpu().numpy(), float(y2.double().sum()), float(y2.double().abs().sum())


def test_learn_step_equals_the_reference():
    g, loss, q, s, a = _learn_step("cpu")
    assert abs(loss - g["loss"]) < 1e-5 and np.allclose(q, g["q_pred"], atol=1e-5)
    assert abs(s - g["after_sum"]) < 1e-3 * abs(g["after_sum"]) and abs(a - g["after_abs_sum"]) < 1e-3 * g["after_abs_sum"]


@pytest.mark.gpu
@pytest.mark.parametrize("tag,make", [("MULTIDISCRETE_RESNET_6", lambda: qnet.MULTIDISCRETE_RESNET(6)), ("RESNET", qnet.RESNET)])
def test_network_equals_the_reference_on_gpu(tag, make):
    """The reference-pinned vectors evaluated on the MI355X (MIOpen convolutions): same weights (seed 0, same layer order), same inputs."""
    g = GOLD[tag]
    torch.manual_seed(0)
    net = make().eval().cuda()
    x = torch.randn(2, 4, 40, 40, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        flat = net(x).reshape(-1).cpu()
    assert np.allclose(flat[torch.tensor(g["sample_idx"])].numpy(), g["sample"], rtol=1e-3, atol=1e-4)
    assert abs(float(flat.double().sum()) - g["sum"]) < 2e-3 * max(1.0, abs(g["sum"]))


@pytest.mark.gpu
def test_learn_step_equals_the_reference_on_gpu():
    g, loss, q, s, a = _learn_step("cuda")
    assert abs(loss - g["loss"]) < 1e-3 and np.allclose(q, g["q_pred"], atol=1e-3)
    assert abs(s - g["after_sum"]) < 1e-2 * abs(g["after_sum"]) and abs(a - g["after_abs_sum"]) < 1e-2 * g["after_abs_sum"]


def _learn_sequence(device, per_round):
    """The golden stream of tools/gen_golden_qnet.py ("learn_sequence": the REFERENCE's ReplayBuffer + network, one push and one learn() per
    transition, Grasping_Agent_multidiscrete.py:551-556) fed to Learner.push_and_learn in rounds of `per_round` transitions."""
    from mujoco_rl_ur5_amd.agent import Learner
    g = GOLD["learn_sequence"]
    n, hs = g["n"], g["size"]
    torch.manual_seed(0)
    net = qnet.MULTIDISCRETE_RESNET(6).train().to(device)
    gen = torch.Generator().manual_seed(4)
    rgb = torch.randint(0, 256, (n, 3, hs, hs), generator=gen).float() / 255.0
    dep = torch.rand(n, 1, hs, hs, generator=gen)
    states = torch.cat((rgb, dep), dim=1).to(device)
    actions = torch.randint(0, 6 * hs * hs, (n, 1), generator=gen).to(device)
    rewards = torch.randint(0, 2, (n, 1), generator=gen).to(device)
    lr = Learner(net, hs, hs, device, mem_size=g["mem"], transitions_per_update=1, max_updates_per_round=10 ** 6)
    losses, ratios = [], []
    for i0 in range(0, n, per_round):
        ls, utd = lr.push_and_learn(states[i0:i0 + per_round], actions[i0:i0 + per_round], rewards[i0:i0 + per_round])
        losses += ls
        ratios.append(utd)
    net.eval()
    with torch.no_grad():
        y = net(states[:2])
    return g, losses, ratios, float(y.double().sum()), float(y.double().abs().sum())


@pytest.mark.parametrize("per_round", [1, 6, 36])
def test_batched_push_and_learn_reproduces_the_reference_s_cadence(per_round):
    """N pushes + N optimiser steps of a batched round == N sequential push / learn() pairs of the reference (update-to-data ratio 1), whatever
    the round size: same sampled batches (python `random` seeded 20, newest transition always in), same losses, same weights afterwards."""
    g, losses, ratios, s, a = _learn_sequence("cpu", per_round)
    assert len(losses) == len(g["losses"]) == g["n"] - 23 and np.allclose(losses, g["losses"], rtol=2e-4, atol=1e-6)
    assert abs(s - g["after_sum"]) < 1e-3 * abs(g["after_sum"]) and abs(a - g["after_abs_sum"]) < 1e-3 * g["after_abs_sum"]
    # every transition pushed once the buffer holds 2 * BATCH_SIZE was followed by a step (13 of the 36; the last rounds run at ratio 1)
    assert ratios[-1] == (1.0 if per_round < 36 else 13 / 36)


def test_update_cap_spreads_the_steps_over_the_round():
    from mujoco_rl_ur5_amd.agent import Learner
    net = qnet.MULTIDISCRETE_RESNET(6)
    lr = Learner(net, 8, 8, "cpu", mem_size=64, max_updates_per_round=4)
    x = torch.rand(40, 4, 8, 8)
    ls, utd = lr.push_and_learn(x, torch.zeros(40, 1, dtype=torch.long), torch.zeros(40, 1))
    # 40 transitions, at most 4 steps: chunks of 10; the first two chunks are below 2 * BATCH_SIZE stored transitions (:396-398)
    assert len(ls) == 2 and utd == 2 / 40 and len(lr.memory) == 40 and lr.updates_done == 2


@pytest.mark.gpu
def test_batched_push_and_learn_reproduces_the_reference_s_cadence_on_gpu():
    """The same stream on the MI355X (MIOpen convolutions, fp32): the first steps agree closely, later ones drift as 13 Adam steps amplify the
    different summation orders of the convolutions -- measured 0 / 0.1 / 2.2 % on the first three and up to 16 % on the last (the 1x1 layers run as GEMMs on the GPU); the sampled batches (python `random`)
    are the same by construction."""
    g, losses, ratios, s, a = _learn_sequence("cuda", 12)
    ref = np.array(g["losses"])
    rel = np.abs(np.array(losses) - ref) / np.maximum(np.abs(ref), 1e-3)
    assert len(losses) == len(ref) and rel[0] < 1e-4 and rel[:3].max() < 5e-2 and rel.max() < 0.3, (rel.round(4).tolist(), s, g["after_sum"], a, g["after_abs_sum"])
    assert abs(a - g["after_abs_sum"]) < 0.1 * g["after_abs_sum"], (a, g["after_abs_sum"])


def _gemm_paths_equal_the_convolutions(device):
    """qnet routes the 1x1 convolutions and the 4-channel first 3x3 convolution through GEMMs on the GPU (MIOpen's immediate mode has only its naive kernels for
    them): outputs, input gradients and weight / bias gradients must equal the nn.Conv2d they replace."""
    import mujoco_rl_ur5_amd.qnet as Q
    torch.manual_seed(3)
    cases = [(torch.nn.Conv2d(64, 128, kernel_size=1, stride=1), Q._conv1x1_as_gemm, (3, 64, 10, 12)),        # BasicBlock.conv3 (with bias)
             (torch.nn.Conv2d(64, 6, kernel_size=1), Q._conv1x1_as_gemm, (2, 64, 16, 16)),                    # the head's C1
             (Q._conv3x3(4, 64), Q._conv3x3_as_gemm, (2, 4, 20, 24))]                                          # Perception_Module.C1
    old = Q._FORCE_GEMM_1X1
    Q._FORCE_GEMM_1X1 = True
    try:
        for conv, fn, shape in cases:
            conv = conv.to(device)
            x1 = torch.randn(shape, device=device, requires_grad=True)
            x2 = x1.detach().clone().requires_grad_(True)
            g = torch.randn(conv(x1).shape, device=device)
            y1 = conv(x1)
            y1.backward(g)
            gw1, gb1 = conv.weight.grad.clone(), None if conv.bias is None else conv.bias.grad.clone()
            conv.zero_grad()
            y2 = fn(conv, x2)
            y2.backward(g)
            assert torch.allclose(y1, y2, rtol=1e-4, atol=1e-5) and torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-5)
            assert torch.allclose(gw1, conv.weight.grad, rtol=1e-4, atol=1e-4) and (gb1 is None or torch.allclose(gb1, conv.bias.grad, rtol=1e-4, atol=1e-4))
        with pytest.raises(AssertionError):
            Q._conv1x1_as_gemm(torch.nn.Conv2d(8, 8, kernel_size=3, padding=1), torch.zeros(1, 8, 4, 4))   # the helper refuses what it does not implement
    finally:
        Q._FORCE_GEMM_1X1 = old


def test_gemm_paths_equal_the_convolutions_on_cpu():
    _gemm_paths_equal_the_convolutions("cpu")


@pytest.mark.gpu
def test_gemm_paths_equal_the_convolutions_on_gpu():
    _gemm_paths_equal_the_convolutions("cuda")
