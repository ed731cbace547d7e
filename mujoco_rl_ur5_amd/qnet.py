"""Pixel-wise grasp-Q network and replay buffer of the reference, for the device-resident rollout loop.

Architecture = ``Modules.py`` of the reference: ``Perception_Module`` (:159-193) + ``Grasping_Module_multidiscrete`` (:243-287),
assembled by ``MULTIDISCRETE_RESNET`` (:308-311); ``RESNET`` / ``POLICY_RESNET`` (:300-305) are the single-map heads. Module and
parameter names are the reference's, so its checkpoints (``checkpoint["model_state_dict"]``) load unchanged, and layers are created
in the reference's order, so a given ``torch.manual_seed`` yields the same initial weights (``tests/test_qnet.py`` checks both
against vectors produced by the reference's own file, ``tools/gen_golden_qnet.py``).

This is plain PyTorch-ROCm (MIOpen convolutions), as BASELINE.json's north_star prescribes for the CNN; what this module adds is
that everything stays on the GPU that simulates: the raster writes straight into torch tensors, the replay buffer is a set of
preallocated device tensors, and one forward pass serves all scenes of the rank.
"""
from __future__ import annotations

import random

import torch
import torch.nn as nn


_FORCE_GEMM_1X1 = False          # tests: take the GEMM path on the host too


class _BatchNorm2d(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` (same parameters, buffers and state_dict keys) with a second training-mode behaviour: inside ``per_sample_statistics()``
    every image is normalised with ITS OWN mean / variance, and the running statistics are left alone.

    Why: the reference never calls ``eval()``; its ``select_action`` (Grasping_Agent_multidiscrete.py:232-299) feeds ONE observation through the
    network in training mode, so batch norm there normalises with that image's statistics. Batching N scenes through the module as is would mix
    the scenes' statistics -- a scene's Q map, and its greedy action, would depend on which other scenes share its chunk (and on how scenes are
    sharded over ranks). Per-image statistics are the exact batched equivalent of N batch-1 training-mode forwards."""
    per_sample = False

    def forward(self, x):
        if not (_BatchNorm2d.per_sample and self.training):
            return super().forward(x)
        n, c, h, w = x.shape
        y = nn.functional.batch_norm(x.reshape(1, n * c, h, w), None, None, self.weight.repeat(n), self.bias.repeat(n), True, 0.0, self.eps)
        return y.view(n, c, h, w)


class per_sample_statistics:
    """Context manager: batch norm layers of this module normalise every image by itself (action selection over a batch of scenes)."""

    def __enter__(self):
        self._old, _BatchNorm2d.per_sample = _BatchNorm2d.per_sample, True

    def __exit__(self, *exc):
        _BatchNorm2d.per_sample = self._old


def _conv1x1_as_gemm(conv, x):
    """A 1x1 convolution as one batched GEMM over the channel axis (rocBLAS / hipBLASLt): [O, C] x [N, C, H*W] -> [N, O, H*W]. Same parameters as the
    ``nn.Conv2d`` it replaces (checkpoints load unchanged). On the MI355X box MIOpen's immediate mode has no tuned solution for these layers and falls
    back to its naive kernel (``naive_conv_ab_nonpacked_*``: 39 % of the DQN loop's GPU time, profiles/r03_m_dqn_kernel_stats.csv)."""
    assert conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1, conv
    if not (x.is_cuda or _FORCE_GEMM_1X1):   # on the host the reference's own operator (the CPU tests compare a 13-step Adam sequence with the reference to 2e-4)
        return conv(x)
    n, c, h, w = x.shape
    y = torch.matmul(conv.weight.view(conv.out_channels, c), x.reshape(n, c, h * w))
    if conv.bias is not None:
        y = y + conv.bias.view(1, -1, 1)
    return y.view(n, conv.out_channels, h, w)


def _conv3x3_as_gemm(conv, x):
    """The network's FIRST convolution (4 -> 64 channels, 3x3, 200 x 200 images) as im2col + one batched GEMM: [64, 36] x [N, 36, H*W]. With 4 input channels
    MIOpen has no Winograd / implicit-GEMM solution and its immediate mode falls back to `naive_conv_ab_nonpacked_{fwd,wrw}` (profiles/r03_m_dqn_kernel_stats.csv);
    the column matrix of a 4-channel image is only 36 rows, and the weight gradient becomes a GEMM as well (the input needs no gradient). Same parameters as the
    ``nn.Conv2d`` it replaces; the host path keeps the reference's operator."""
    assert conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None, conv
    if not (x.is_cuda or _FORCE_GEMM_1X1):
        return conv(x)
    n, c, h, w = x.shape
    cols = nn.functional.unfold(x, kernel_size=3, padding=1)                          # [N, 9 C, H W], rows ordered (channel, ky, kx) like weight.view(O, -1)
    return torch.matmul(conv.weight.view(conv.out_channels, 9 * c), cols).view(n, conv.out_channels, h, w)


def _conv3x3(cin, cout):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=False)


class BasicBlock(nn.Module):
    """Modules.py:92-142 -- two 3x3 convolutions with batch norm; a 1x1 convolution (with bias) on the skip path when the channel
    count changes. No stride, no down-sampling in this network."""
    expansion = 1

    def __init__(self, inplanes, planes):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes)
        self.bn1 = _BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = _BatchNorm2d(planes)
        self.downsample = None
        self.stride = 1
        self.conv3 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=1) if inplanes != planes else None

    def forward(self, x):
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        skip = x if self.conv3 is None else _conv1x1_as_gemm(self.conv3, x)
        return self.relu(out + skip)


class Perception_Module(nn.Module):
    """Modules.py:159-193 -- 4-channel RGB-D in, 512 channels at 1/4 resolution out."""

    def __init__(self):
        super().__init__()
        self.C1 = _conv3x3(4, 64)
        self.MP1 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.RB1 = BasicBlock(64, 128)
        self.MP2 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.RB2 = BasicBlock(128, 256)
        self.RB3 = BasicBlock(256, 512)

    def forward(self, x, verbose=0):
        return self.RB3(self.RB2(self.MP2(self.RB1(self.MP1(_conv3x3_as_gemm(self.C1, x))))))


class _GraspingHead(nn.Module):
    def __init__(self, out_channels, output_activation):
        super().__init__()
        self.RB1 = BasicBlock(512, 256)
        self.RB2 = BasicBlock(256, 128)
        self.UP1 = nn.UpsamplingBilinear2d(scale_factor=2)
        self.RB3 = BasicBlock(128, 64)
        self.UP2 = nn.UpsamplingBilinear2d(scale_factor=2)
        self.C1 = nn.Conv2d(64, out_channels, kernel_size=1)
        self.output_activation = output_activation
        if output_activation is not None:
            self.sigmoid = nn.Sigmoid()

    def forward(self, x, verbose=0):
        x = _conv1x1_as_gemm(self.C1, self.UP2(self.RB3(self.UP1(self.RB2(self.RB1(x))))))
        x = x.squeeze()                                   # the reference squeezes in place (:230,:277): batch 1 loses its batch axis
        return self.sigmoid(x) if self.output_activation is not None else x


class Grasping_Module(_GraspingHead):
    """Modules.py:196-240 -- one Q map."""

    def __init__(self, output_activation="Sigmoid"):
        super().__init__(1, output_activation)


class Grasping_Module_multidiscrete(_GraspingHead):
    """Modules.py:243-287 -- one Q map per wrist rotation."""

    def __init__(self, output_activation="Sigmoid", act_dim_2=6):
        super().__init__(act_dim_2, output_activation)


def RESNET():
    return nn.Sequential(Perception_Module(), Grasping_Module())


def POLICY_RESNET():
    return nn.Sequential(Perception_Module(), Grasping_Module(output_activation=None))


def MULTIDISCRETE_RESNET(number_actions_dim_2):
    return nn.Sequential(Perception_Module(), Grasping_Module_multidiscrete(act_dim_2=number_actions_dim_2))


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


class ReplayBuffer:
    """``ReplayBuffer(size, simple=True)`` of Modules.py:28-55 as preallocated tensors on ``device``: ring overwrite, ``sample(b)`` =
    b-1 transitions drawn without replacement plus the most recent one (:46-49), python ``random`` seeded with 20 (:33).

    Batched use: ``push`` takes N transitions at once (one per scene of the rank) and stores them in scene order, which is what N
    consecutive pushes of the reference would do. States are kept as uint8 RGB + float32 normalised depth (7 bytes per pixel instead of
    the reference's 16; the 1e-3 depth noise of transform_observation survives, which float16's 5e-4 steps would not guarantee) and expanded to the 4-channel float tensor on ``sample``."""

    def __init__(self, size, height=200, width=200, device="cpu", simple=True, seed=20):
        if not simple:
            raise NotImplementedError("GAMMA = 0 in the reference (Grasping_Agent_multidiscrete.py:32): only simple transitions are stored")
        self.size, self.position, self.count = int(size), 0, 0
        self.device = torch.device(device)
        self.rgb = torch.zeros((self.size, 3, height, width), dtype=torch.uint8, device=self.device)
        self.depth = torch.zeros((self.size, 1, height, width), dtype=torch.float32, device=self.device)
        self.action = torch.zeros((self.size, 1), dtype=torch.long, device=self.device)
        self.reward = torch.zeros((self.size, 1), dtype=torch.float32, device=self.device)
        self._rng = random.Random(seed)

    def __len__(self):
        return self.count

    def push(self, state, action, reward):
        """state [N,4,H,W] float in [0,1] (rgb/255, normalised depth), action [N] or [N,1] long, reward [N] or [N,1]."""
        n = state.shape[0]
        if n > self.size:   # N consecutive pushes into a ring of `size` keep the last `size`: drop the rest up front (no duplicate indices)
            drop = n - self.size
            state, action, reward = state[drop:], action.reshape(n, -1)[drop:], reward.reshape(n, -1)[drop:]
            self.position = (self.position + drop) % self.size
            n = self.size
        idx = (self.position + torch.arange(n, device=self.device)) % self.size
        self.rgb[idx] = (state[:, :3].to(self.device) * 255.0).round().clamp(0, 255).to(torch.uint8)
        self.depth[idx] = state[:, 3:4].to(self.device).float()
        self.action[idx] = action.reshape(n, 1).to(self.device).long()
        self.reward[idx] = reward.reshape(n, 1).to(self.device).float()
        self.position = (self.position + n) % self.size
        self.count = min(self.size, self.count + n)

    def sample(self, batch_size):
        if self.count < batch_size:
            raise ValueError("not enough transitions")
        last = (self.position - 1) % self.size
        picks = self._rng.sample(range(self.count), batch_size - 1) + [last]
        idx = torch.tensor(picks, device=self.device)
        state = torch.cat((self.rgb[idx].float() / 255.0, self.depth[idx].float()), dim=1)
        if self.shared:
            # every rank drew the same slots (same python RNG, same ring bookkeeping); a slot's image lives on the rank that simulated the scene.
            # Each rank contributes the rows it owns, zeros elsewhere; the sum over ranks is the batch -- exact in fp32 (x + 0 + ... + 0).
            import torch.distributed as dist
            state = state * self.owned[idx].view(-1, 1, 1, 1).to(state.dtype)
            if dist.get_backend() != "nccl" and state.is_cuda:      # gloo with several ranks on one GPU (tests): reduce through host memory
                host = state.cpu()
                dist.all_reduce(host)
                state = host.to(self.device)
            else:
                from . import sharding
                sharding._timed("all_reduce_replay_batch", state.numel() * state.element_size(), lambda: dist.all_reduce(state), state.is_cuda)
        return state, self.action[idx], self.reward[idx]

    # ---- the SHARED replay buffer of a multi-rank job (BASELINE.json north_star: "RCCL ... to all-gather grasp outcomes into the shared replay buffer")
    # One logical ring for the whole job, in GLOBAL scene order -- exactly the ring a single process simulating all scenes would fill
    # (Grasping_Agent_multidiscrete.py:551-554 is the single push site). Its bookkeeping (position, count, action, reward: what the gathered 16-byte
    # outcome records carry) is replicated on every rank; the 280 KB observation of a transition stays on the rank that rendered it (`owned`).
    shared = False

    def make_shared(self):
        self.shared = True
        self.owned = torch.zeros(self.size, dtype=torch.bool, device=self.device)
        return self

    def push_shared(self, gid0, actions, rewards, local_state, local_lo):
        """Transitions of the global scenes gid0 .. gid0 + len(actions) - 1 (consecutive ids): actions / rewards of ALL of them (from the gathered
        outcome records), and the states of the ones this rank simulated -- ``local_state[i]`` belongs to global scene ``local_lo + i``."""
        n = int(actions.shape[0])
        drop = max(0, n - self.size)                    # as in push(): n consecutive pushes keep the last `size`
        if drop:
            self.position = (self.position + drop) % self.size
        idx = (self.position + torch.arange(n - drop, device=self.device)) % self.size
        self.action[idx] = actions[drop:].reshape(-1, 1).to(self.device).long()
        self.reward[idx] = rewards[drop:].reshape(-1, 1).to(self.device).float()
        # the scenes of the chunk this rank simulated: an interval of consecutive ids, computed on the host (gid0, n, local_lo are python ints: no
        # device-to-host synchronisation per chunk -- round-4 advice: `bool(mine.any())` stalled the stream up to 64 times per round)
        g_lo, g_hi = max(gid0 + drop, int(local_lo)), min(gid0 + n, int(local_lo) + int(local_state.shape[0]))
        self.owned[idx] = False
        if g_hi > g_lo:
            sel = idx[g_lo - (gid0 + drop):g_hi - (gid0 + drop)]
            st = local_state[g_lo - int(local_lo):g_hi - int(local_lo)]
            self.owned[sel] = True
            self.rgb[sel] = (st[:, :3].to(self.device) * 255.0).round().clamp(0, 255).to(torch.uint8)
            self.depth[sel] = st[:, 3:4].to(self.device).float()
        self.position = (self.position + n - drop) % self.size
        self.count = min(self.size, self.count + n)
