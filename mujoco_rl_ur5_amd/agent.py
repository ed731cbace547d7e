"""Batched, device-resident version of the reference's DQN grasp agent (BASELINE.json config 5).

Mirrors ``Grasp_Agent`` of ``Grasping_Agent_multidiscrete.py`` (:47-446): ``transform_observation`` (:301-368), ``epsilon_greedy``
(:232-282) incl. the "resample until the pixel lies on the table" rule (:266-280), ``transform_action`` (:380-385), ``learn`` (:388-446:
gamma = 0, binary cross entropy between Q(s, a) and the grasp outcome, Adam lr 1e-3, weight decay 2e-5, batch 12, the newest
transition always in the batch) -- for N scenes per rank at once. Observations are rendered by the engine into torch tensors on the
simulating GPU, the network runs there (PyTorch-ROCm / MIOpen), actions go back to the engine as a device tensor, rewards come back
as one; per round only the 16-byte outcome records cross ranks (``sharding.gather_outcomes``, RCCL all_gather).

Several ranks = ONE agent (round 4): the job's scenes shard over the ranks, every random draw is keyed by global scene id and round
(``sharding.scene_uniform``), action selection normalises every image by itself (``qnet.per_sample_statistics``), and the gathered outcome
records drive one logical replay ring (``qnet.ReplayBuffer.push_shared``) from which every rank takes the same optimiser steps -- so N ranks
with n scenes each produce the outcome records, the loss sequence and the weights of one rank with N n scenes (``tests/test_sharding.py``). The
images of a sampled batch travel to the replicas in one all-reduce per optimiser step (12 x 640 KB); gradients never do. That equality is EXACT on
the CPU (gloo tests: sha256 of the weights). On GPUs the replicas' optimiser steps differ at rounding level (MIOpen's weight-gradient kernels
accumulate with atomics; ``tests/test_sharding.py`` accepts 15 % on the per-rank losses within a round), so every round ends with one flattened
broadcast of rank 0's weights, batch-norm buffers and Adam state (88 MB, ``sharding.broadcast_many_from_rank0``): the copies are re-identified once
per round, they are not bit-identical in between.

Differences that follow from batching, all deliberate: epsilon decays per transition (``steps_done`` advances by the job's scene count per round); colour
jitter (torchvision's ColorJitter, :120-126) and the depth noise run on the device (``color_jitter`` below). Learning cadence: the reference
pushes ONE transition and takes ONE optimiser step per env step (:551-556). ``Learner.push_and_learn`` keeps that order for a round's N
transitions -- push k, step, push k, step ... -- with ``transitions_per_update = k`` (default 1 = the reference's update-to-data ratio of 1)
and at most ``max_updates_per_round`` steps per round (default 64: a round of 4096 scenes then learns from every 64th push); the round's
output states the ratio it ran at (``update_to_data``).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import sharding
from .envs import GraspEnv
from .qnet import MULTIDISCRETE_RESNET, ReplayBuffer, per_sample_statistics

MEMORY_SIZE = 2000            # Grasping_Agent_multidiscrete.py:26-38
BATCH_SIZE = 12
LEARNING_RATE = 0.001
EPS_START, EPS_END, EPS_DECAY = 1.0, 0.2, 8000


def _gray(img):
    return (0.2989 * img[:, 0] + 0.587 * img[:, 1] + 0.114 * img[:, 2]).unsqueeze(1)


def _rgb_to_hsv(img):
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    maxc, minc = img.max(dim=1).values, img.min(dim=1).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=1)


def _hsv_to_rgb(img):
    h, s, v = img[:, 0], img[:, 1], img[:, 2]
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p, q, t = (v * (1.0 - s)).clamp(0, 1), (v * (1.0 - s * f)).clamp(0, 1), (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
    i = i.long().unsqueeze(1)                                   # [N, 1, H, W]: one gather per channel instead of a 6-way one-hot product
    a1 = torch.stack((v, q, p, p, t, v), dim=1).gather(1, i)
    a2 = torch.stack((t, v, v, q, p, p), dim=1).gather(1, i)
    a3 = torch.stack((p, p, t, v, v, q), dim=1).gather(1, i)
    return torch.cat((a1, a2, a3), dim=1)


def color_jitter(rgb, generator=None, brightness=0.5, contrast=0.5, saturation=0.5, hue=0.5, u=None):
    """torchvision ``T.ColorJitter(brightness=0.5, contrast=0.5, saturation=0.5, hue=0.5)`` (Grasping_Agent_multidiscrete.py:120-126) for a
    whole batch ON THE DEVICE: rgb float [N, 3, H, W] in [0, 1]. Every image draws its own four factors -- brightness / contrast / saturation
    from U(1 - x, 1 + x), hue from U(-x, x) -- and its own order of the four operations, as torchvision's forward() does per call; the
    operations are torchvision's tensor definitions (blend with black / mean grey / grey image, hue through HSV). The reference goes through
    a uint8 PIL image between ToPILImage and ToTensor; this stays in float."""
    n, dev = rgb.shape[0], rgb.device
    if u is None:                                              # u: the 8 uniform draws of every image, given by callers that key them by scene id
        u = torch.rand((n, 8), device=dev, generator=generator)
    fb = (1 - brightness + 2 * brightness * u[:, 0]).view(n, 1, 1, 1)
    fc = (1 - contrast + 2 * contrast * u[:, 1]).view(n, 1, 1, 1)
    fs = (1 - saturation + 2 * saturation * u[:, 2]).view(n, 1, 1, 1)
    fh = (-hue + 2 * hue * u[:, 3]).view(n, 1, 1)
    order = torch.argsort(u[:, 4:8], dim=1)                    # a random permutation of the 4 ops per image
    out = rgb.clone()
    for pos in range(4):                                       # each operation runs ONCE per position, on the images whose order selects it there
        for op in range(4):
            idx = (order[:, pos] == op).nonzero(as_tuple=True)[0]
            if idx.numel() == 0:
                continue
            x = out.index_select(0, idx)
            if op == 0:
                y = (x * fb[idx]).clamp(0, 1)
            elif op == 1:
                y = (fc[idx] * x + (1 - fc[idx]) * _gray(x).mean(dim=(1, 2, 3), keepdim=True)).clamp(0, 1)
            elif op == 2:
                y = (fs[idx] * x + (1 - fs[idx]) * _gray(x)).clamp(0, 1)
            else:
                hsv = _rgb_to_hsv(x)
                y = _hsv_to_rgb(torch.stack((torch.remainder(hsv[:, 0] + fh[idx], 1.0), hsv[:, 1], hsv[:, 2]), dim=1))
            out.index_copy_(0, idx, y)
    return out


class Learner:
    """Replay buffer + optimiser step of ``Grasp_Agent`` (Grasping_Agent_multidiscrete.py:140-156, 388-446, 551-556) without the environment:
    what ``tests/test_qnet.py`` replays against a transition stream run through the reference's own ``Modules.ReplayBuffer`` and network."""

    def __init__(self, policy_net, height, width, device, learning_rate=LEARNING_RATE, mem_size=MEMORY_SIZE, batch_size=BATCH_SIZE,
                 transitions_per_update=1, max_updates_per_round=64):
        self.policy_net, self.device, self.batch_size = policy_net, torch.device(device), int(batch_size)
        self.memory = ReplayBuffer(mem_size, height, width, device=self.device)                          # :140-142
        self.optimizer = torch.optim.Adam(self.policy_net.parameters(), lr=learning_rate, weight_decay=0.00002)   # :153-156
        self.transitions_per_update, self.max_updates_per_round = max(1, int(transitions_per_update)), max(1, int(max_updates_per_round))
        self.last_loss, self.updates_done = None, 0

    def learn(self):
        """:388-446 with GAMMA = 0: one optimiser step on BATCH_SIZE transitions (the newest always included)."""
        if len(self.memory) < 2 * self.batch_size:                                                        # :396-398
            return None
        state, action, reward = self.memory.sample(self.batch_size)
        self.policy_net.train()
        q_pred = self.policy_net(state).reshape(self.batch_size, -1).gather(1, action)                   # :424-426
        loss = F.binary_cross_entropy(q_pred, reward.float())                                            # :439
        loss.backward()
        self.optimizer.step()
        self.optimizer.zero_grad()
        self.last_loss = float(loss.detach())
        self.updates_done += 1
        return self.last_loss

    def feed_round(self, n, parts, learn=True):
        """push_and_learn for a round whose N transitions arrive in PARTS (scene order): ``parts`` yields (state, action, reward) of consecutive scene ranges as
        they become available (a generator that waits for a scene group's physics to finish). The chunking is the round's -- k consecutive scenes per optimiser
        step, k from the whole round's N --, a chunk that straddles two parts waits for the later one: the pushes, batches and steps are exactly those of
        push_and_learn on the concatenated round, they only start earlier (while the next group's grasp launch is still running)."""
        k = max(self.transitions_per_update, -(-n // self.max_updates_per_round))
        losses, held = [], None
        for st, ac, rw in parts:
            if held is not None:
                st, ac, rw = torch.cat((held[0], st)), torch.cat((held[1], ac)), torch.cat((held[2], rw))
            m = (st.shape[0] // k) * k
            for i0 in range(0, m, k):
                self.memory.push(st[i0:i0 + k], ac[i0:i0 + k], rw[i0:i0 + k])
                if learn:
                    loss = self.learn()
                    if loss is not None:
                        losses.append(loss)
            held = (st[m:], ac[m:], rw[m:]) if m < st.shape[0] else None
        if held is not None:                                                                              # the round's last, shorter chunk
            self.memory.push(*held)
            if learn:
                loss = self.learn()
                if loss is not None:
                    losses.append(loss)
        return losses, (len(losses) / n if n else 0.0)

    def push_and_learn(self, state, action, reward, learn=True, outcomes=None, first_scene_id=0, n_actions_1=None):
        """A round's N transitions in the reference's order (:551-556: push, then learn): the transitions go into the ring in chunks of
        ``k`` consecutive scenes, an optimiser step after each chunk (its newest transition is in the batch). k = transitions_per_update,
        raised so that a round takes at most max_updates_per_round steps. Returns (losses, update_to_data ratio of this round).

        Multi-rank jobs pass ``outcomes`` -- the all-gathered [n_total, 4] records {scene id, pixel, rotation, reward} of the round -- and the id
        of this rank's first scene: the round's n_total transitions then go through ONE logical ring in global scene order
        (``ReplayBuffer.push_shared``), every rank takes the SAME optimiser steps on the same batches (``sample`` assembles a batch from the
        ranks that hold its images) without any gradient traffic: one learner, as in the reference, fed by all the shards. The replicas of
        ``policy_net`` stay bit-identical on the CPU; on GPUs they agree to rounding within a round and are re-identified by the per-round broadcast
        of ``BatchedGraspAgent.round`` (module docstring)."""
        shared = outcomes is not None and self.memory.shared
        n = int(outcomes.shape[0]) if shared else state.shape[0]
        k = max(self.transitions_per_update, -(-n // self.max_updates_per_round))
        losses = []
        for i0 in range(0, n, k):
            if shared:
                o = outcomes[i0:i0 + k].to(self.device).long()
                self.memory.push_shared(i0, o[:, 2] * n_actions_1 + o[:, 1], o[:, 3], state, first_scene_id)
            else:
                self.memory.push(state[i0:i0 + k], action[i0:i0 + k], reward[i0:i0 + k])
            if learn:
                loss = self.learn()
                if loss is not None:
                    losses.append(loss)
        return losses, (len(losses) / n if n else 0.0)


class BatchedGraspAgent:
    def __init__(self, env: GraspEnv = None, n_envs=1, device=None, learning_rate=LEARNING_RATE, mem_size=MEMORY_SIZE, eps_start=EPS_START,
                 eps_end=EPS_END, eps_decay=EPS_DECAY, seed=20, load_path=None, first_scene_id=0, n_total=None, transitions_per_update=1,
                 max_updates_per_round=64, pipeline_groups=1, **env_kwargs):
        torch.manual_seed(seed)                                                        # :76-79
        np.random.seed(seed)
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        # pipeline_groups = G > 1 (round 5): the rank's scenes are G scene groups -- one GraspEnv (engine handle) + one CUDA stream each, consecutive scene ranges.
        # A round queues group 0's render -> CNN forward -> action selection -> grasp launch, then does the same for group 1 WHILE group 0's grasp launch occupies
        # the chip, and so on; the replay pushes and optimiser steps of a group start as soon as its rewards are in, under the next group's launch. Every group's
        # actions are selected with the weights the round started with, pushes keep scene order and the round's chunking: the transitions, batches and optimiser steps
        # are those of pipeline_groups = 1 (tests/test_agent.py), only the engine no longer idles through the CNN and the learner.
        self.G = max(1, int(pipeline_groups))
        if env is not None:
            if self.G > 1:
                raise ValueError("pipeline_groups > 1 builds its own scene groups: pass n_envs / first_scene_id / n_total instead of env")
            self.envs, self.streams = [env], None
        else:
            if n_envs % self.G:
                raise ValueError("n_envs must be a multiple of pipeline_groups")
            ng = n_envs // self.G
            if n_total is None and self.G > 1:                                          # the groups' envs cannot derive the job's scene count from their own ranges
                import torch.distributed as dist
                world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
                if first_scene_id != (dist.get_rank() if world > 1 else 0) * n_envs:
                    raise ValueError("first_scene_id > 0 needs n_total (the global scene count)")
                n_total = world * n_envs
            kw = dict(dict(show_obs=False, observation="render"), **env_kwargs)
            self.envs, self.streams = [], None
            for g in range(self.G):
                self.envs.append(GraspEnv(n_envs=ng, first_scene_id=first_scene_id + g * ng, n_total=n_total, **kw))
            if self.G > 1 and self.device.type == "cuda":
                # one CUDA stream per group, on DIFFERENT hardware queues -- chosen by measurement (streams.py): which queue the HIP runtime gives a stream depends on the
                # process's whole stream history, and two group streams on one queue serialise the groups (profiles/r05_f_dqn512_timeline_one_queue.txt)
                from .streams import group_streams
                self.streams, self.streams_overlap_verified = group_streams(torch, self.device, self.G)
                for e, st in zip(self.envs, self.streams):
                    e.use_stream(st)
        self.env = self.envs[0]                                                         # model constants, action space, camera (equal in every group)
        self.N, self.H, self.W = sum(e.n_envs for e in self.envs), self.env.IMAGE_HEIGHT, self.env.IMAGE_WIDTH
        self.n_actions_1, self.n_actions_2 = int(self.env.action_space.nvec[0]), int(self.env.action_space.nvec[1])   # :97-100
        self.output = self.n_actions_1 * self.n_actions_2
        self.policy_net = MULTIDISCRETE_RESNET(number_actions_dim_2=self.n_actions_2).to(self.device)     # :103
        checkpoint = None
        if load_path is not None:                                                      # :109-114
            checkpoint = torch.load(load_path, map_location=self.device)
            self.policy_net.load_state_dict(checkpoint["model_state_dict"])
        self.depth_threshold = float(np.round(self.env.model.cam_pos0[self.env.model.camera_name2id("top_down")][2]
                                              - self.env.TABLE_HEIGHT + 0.01, decimals=3))   # :131-136
        self.learner = Learner(self.policy_net, self.H, self.W, self.device, learning_rate, mem_size, BATCH_SIZE, transitions_per_update,
                               max_updates_per_round)                                  # :140-156
        self.memory, self.optimizer = self.learner.memory, self.learner.optimizer
        self.eps_start, self.eps_end, self.eps_decay = eps_start, eps_end, eps_decay
        self.steps_done, self.eps_threshold = 0, eps_start
        # :165-180 / :215-217 / :448-465: how often the greedy policy chose each rotation, and the successes per rotation of greedy and of random actions. Kept as device
        # counters [3, n_rotations] (no host read per round); the reference's three dicts are the properties below and what save() writes.
        self._rot_counts = torch.zeros((3, self.n_actions_2), dtype=torch.int64, device=self.device)
        if checkpoint is not None and "optimizer_state_dict" in checkpoint:            # :157-180: a checkpoint of save() / of the reference's trainer resumes the whole trainer
            self.optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
            self.steps_done = int(checkpoint.get("step", 0))
            self.eps_threshold = float(checkpoint.get("epsilon", eps_end))
            for row, key in enumerate(("greedy_rotations", "greedy_rotations_successes", "random_rotations_successes")):
                for rot, cnt in dict(checkpoint.get(key, {})).items():
                    if 0 <= int(rot) < self.n_actions_2:
                        self._rot_counts[row, int(rot)] = int(cnt)
        self.first_scene_id = self.env.first_scene_id                                   # one source of truth: the env's scene range
        self.n_total = self.env.n_total

        self.last_loss = None
        # every random draw of the loop is keyed by (seed, GLOBAL scene id, round): a scene explores, jitters and is noised the same way however the
        # batch is sharded (sharding.scene_uniform)
        self.seed, self.rounds_done = int(seed), 0
        self.gids = self.first_scene_id + torch.arange(self.N, dtype=torch.int64, device=self.device)
        import torch.distributed as dist
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # several ranks -- or one rank told to take the collective path anyway (sharding.FORCE_COLLECTIVES: how RCCL is exercised on a one-GPU box)
        self.shared = sharding.collectives_active()
        if self.shared:
            if self.n_total != self.world * self.N or self.first_scene_id != dist.get_rank() * self.N:
                raise ValueError("multi-rank agent: every rank simulates the contiguous shard sharding.shard_range(n_total, rank, world)")
            self.memory.make_shared()                                                   # one logical replay ring for the job (qnet.ReplayBuffer.push_shared)
            sharding.broadcast_many_from_rank0([t.data for t in list(self.policy_net.parameters()) + list(self.policy_net.buffers())])   # one set of initial weights (29 MB, once): rank 0's

    # ------------------------------------------------------------------ observation -> network input
    def transform_observation(self, observation, normalize=True, jitter_and_noise=True, gids=None):
        """:301-368 for a batch: depth clipped at the table threshold, (noise,) negated and min-max normalised per image; rgb / 255.
        observation = {"rgb": uint8 [N,H,W,3], "depth": float32 [N,H,W]} device tensors -> float32 [N,4,H,W]. gids: global ids of the scenes shown (default: all)."""
        gids = self.gids if gids is None else gids
        n = int(gids.shape[0])
        depth = observation["depth"].to(self.device).float().clamp(max=self.depth_threshold)            # :311
        if normalize:
            for i0 in range(0, n, 128):                                                                  # :317 (whenever normalize=True); chunks bound the int64 temporaries (128 x 80 000 x 8 B = 82 MB each)
                depth[i0:i0 + 128] += 0.001 * sharding.scene_normal(self.seed, gids[i0:i0 + 128], self.rounds_done, 4, self.H * self.W).view(-1, self.H, self.W)
            depth = -depth
            dmin = depth.amin(dim=(1, 2), keepdim=True)
            dmax = depth.amax(dim=(1, 2), keepdim=True)
            depth = (depth - dmin) / (dmax - dmin).clamp_min(1e-12)                                      # :319-321
        rgb = observation["rgb"].to(self.device).permute(0, 3, 1, 2).float() / 255.0                    # ToTensor (:128)
        if normalize and jitter_and_noise:
            rgb = color_jitter(rgb, u=sharding.scene_uniform(self.seed, gids, self.rounds_done, 5, 8))   # self.normal_rgb (:117-123, :334-335)
        return torch.cat((rgb, depth.unsqueeze(1)), dim=1)

    # ------------------------------------------------------------------ action selection
    def begin_round_epsilon(self):
        """:241-243, once per round: every scene of the round -- whichever group selects it -- draws against the same threshold."""
        self.eps_threshold = self.eps_end + (self.eps_start - self.eps_end) * math.exp(-1.0 * self.steps_done / self.eps_decay)   # :241-243
        self.steps_done += self.n_total                                                                  # epsilon decays per transition of the whole job

    def epsilon_greedy(self, state, observation, gids=None, env=None, new_round=True):
        """:232-282 per scene: greedy = argmax over the [6, H, W] Q maps; random = uniform over the (pixel, rotation) pairs whose pixel
        lies on the table (world z >= TABLE_HEIGHT - 0.01, :266-280). Returns (action long [N], greedy bool [N]). gids / env: a scene group of the round."""
        if new_round:
            self.begin_round_epsilon()
        gids = self.gids if gids is None else gids
        env = self.env if env is None else env
        n = int(gids.shape[0])
        u = sharding.scene_uniform(self.seed, gids, self.rounds_done, 1, 3, dtype=torch.float64)         # explore?, which table pixel, which rotation
        explore = u[:, 0] <= self.eps_threshold
        greedy_action = self._q_all(state)[1]
        world = env.pixel_world_device(observation["depth"], self.device)                                # [N,H,W,3]
        on_table = (world[..., 2] >= env.TABLE_HEIGHT - 0.01).reshape(n, -1)
        on_table = torch.where(on_table.any(dim=1, keepdim=True), on_table, torch.ones_like(on_table))
        cdf = on_table.cumsum(dim=1, dtype=torch.int32)                                                  # uniform over the table pixels: inverse CDF of the draw (int32: 40 000 pixels)
        want = torch.floor(u[:, 1] * cdf[:, -1].double()).to(torch.int32) + 1                            # the want-th table pixel, 1-based
        pixel = torch.searchsorted(cdf, want[:, None]).squeeze(1).clamp(max=self.n_actions_1 - 1)
        rot = torch.floor(u[:, 2] * self.n_actions_2).long().clamp(max=self.n_actions_2 - 1)
        random_action = rot * self.n_actions_1 + pixel
        return torch.where(explore, random_action, greedy_action), ~explore

    def _q_all(self, state, chunk=256):
        """(max Q [N], argmax over the 6*H*W (rotation, pixel) pairs [N]) of every scene: the network runs on `chunk` scenes at a time -- its
        first layer alone is 10 MB of activations per 200x200 scene, so thousands of scenes in one call would need tens of GB. Batch norm
        uses every image's OWN statistics here (qnet.per_sample_statistics): the reference selects actions with a batch of one in training
        mode (:232-299), and a scene's greedy action must not depend on which scenes share its chunk or its rank."""
        vals, idxs = [], []
        n = int(state.shape[0])
        with torch.no_grad(), per_sample_statistics():
            for i0 in range(0, n, chunk):
                q = self.policy_net(state[i0:i0 + chunk]).reshape(min(chunk, n - i0), -1)                # [c, 6*H*W]
                v, i = q.max(dim=1)
                vals.append(v)
                idxs.append(i)
        return torch.cat(vals), torch.cat(idxs)

    def greedy(self, state):                                                                             # :284-299
        value, idx = self._q_all(state)
        return idx, value

    def transform_action(self, action):
        """:380-385: flat index -> [pixel, rotation] (the Q maps are laid out [rotation][pixel])."""
        return torch.stack((action % self.n_actions_1, action // self.n_actions_1), dim=1)

    # ------------------------------------------------------------------ learning
    def learn(self):
        """One optimiser step (:388-446), see ``Learner.learn``."""
        self.last_loss = self.learner.learn()
        return self.last_loss

    # ------------------------------------------------------------------ the same round with the scene groups pipelined (pipeline_groups > 1)
    def _round_pipelined(self, learn, return_observation):
        import contextlib
        cuda = self.streams is not None
        main = torch.cuda.current_stream(self.device) if cuda else None
        self.begin_round_epsilon()
        parts, o, selected = [], 0, []
        probe = getattr(self, "_order_probe", None)                                                      # tests/test_agent.py: dict(flag=int tensor, delay_cycles=int, seen=[])
        if cuda and probe is not None:
            probe["flag"].zero_()
        for g, env in enumerate(self.envs):                                                              # queue every group's rollout: the host never waits here
            gids = self.gids[o:o + env.n_envs]
            o += env.n_envs
            if cuda:
                self.streams[g].wait_stream(main)                                                        # this round's weights (the previous round's optimiser steps ran on the main stream)
            with (torch.cuda.stream(self.streams[g]) if cuda else contextlib.nullcontext()):
                if cuda and probe is not None and g > 0:
                    torch.cuda._sleep(int(probe["delay_cycles"]))                                        # test hook: a late group (its CNN would still be running when group 0's launch ends)
                obs = env.observation_device(self.device, sync=not cuda)
                raw = {k: v.clone() for k, v in obs.items()} if return_observation else None
                state = self.transform_observation(obs, gids=gids)
                action, greedy = self.epsilon_greedy(state, obs, gids=gids, env=env, new_round=False)
                env_action = self.transform_action(action)
                if cuda:                                                                                 # group g has READ the round-start weights from here on: the learner's first optimiser
                    selected.append(torch.cuda.Event())                                                  # step (main stream) waits for every group's event, or it would overwrite policy_net
                    selected[-1].record(self.streams[g])                                                 # under a later group's CNN forward (advisor, round 5)
                    if probe is not None:
                        probe["seen"].append(probe["flag"].clone())                                      # 0 = chosen before the learner's first step of this round was queued to run
                reward, skipped = env.step_device(env_action, obs["depth"], self.device, sync=not cuda)   # group g's grasp launch: the next group's CNN runs under it
                rec = torch.stack([gids.int(), env_action[:, 0].int(), env_action[:, 1].int(), reward.int()], dim=1)   # (reward is read when the launch has run: stream order)
            parts.append(dict(raw=raw, state=state, action=action, greedy=greedy, reward=reward, skipped=skipped, rec=rec))

        def finished():                                                                                  # a group's transitions once its launch has run, in scene order
            for ev in selected:
                main.wait_event(ev)                                                                      # every group has chosen its actions with the round-start weights
            if cuda and probe is not None:
                probe["flag"].fill_(1)                                                                   # (main stream, in front of the first optimiser step)
            for g, p in enumerate(parts):
                if cuda:
                    main.wait_stream(self.streams[g])                                                    # the learner's stream waits for group g only; later groups keep running
                yield p["state"], p["action"], p["reward"]
        if self.shared:                                                                                  # several ranks: the ring is filled from the gathered records of ALL ranks, after the round
            for _ in finished():
                pass
            rec = torch.cat([p["rec"] for p in parts])
            outcomes = sharding.gather_outcomes(rec)
            losses, utd = self.learner.push_and_learn(torch.cat([p["state"] for p in parts]), torch.cat([p["action"] for p in parts]), torch.cat([p["reward"] for p in parts]),
                                                      learn=learn, outcomes=outcomes, first_scene_id=self.first_scene_id, n_actions_1=self.n_actions_1)
        else:                                                                                            # one rank: group g's pushes and optimiser steps run under group g + 1's launch
            losses, utd = self.learner.feed_round(self.N, finished(), learn=learn)
            outcomes = sharding.gather_outcomes(torch.cat([p["rec"] for p in parts]))
        for env in self.envs:
            if cuda:
                env.sim.sync()
            env.check_status()
        cat = lambda key: torch.cat([p[key] for p in parts])
        raw = {k: torch.cat([p["raw"][k] for p in parts]) for k in parts[0]["raw"]} if return_observation else None
        return self._end_round(learn, losses, utd, raw, cat("action"), cat("reward").clone(), cat("skipped"), cat("greedy"), outcomes)

    # ------------------------------------------------------------------ one round of the episode loop (:540-560)
    def round(self, learn=True, return_observation=False):
        """observe -> act -> grasp -> store -> learn, for every scene of this rank; outcomes gathered over ranks (X1).
        return_observation: also hand back copies of the raw observation the actions were chosen in (what generate_data.py stores)."""
        if self.G > 1:
            return self._round_pipelined(learn, return_observation)
        obs = self.env.observation_device(self.device)
        raw = {k: v.clone() for k, v in obs.items()} if return_observation else None
        state = self.transform_observation(obs)
        action, greedy = self.epsilon_greedy(state, obs)
        env_action = self.transform_action(action)
        reward, skipped = self.env.step_device(env_action, obs["depth"], self.device)
        rec = torch.stack([self.gids.int(), env_action[:, 0].int(), env_action[:, 1].int(), reward.int()], dim=1)
        outcomes = sharding.gather_outcomes(rec)                                                         # the round's only collective on the rollout side: 16 B per scene
        losses, utd = self.learner.push_and_learn(state, action, reward, learn=learn, outcomes=outcomes if self.shared else None,
                                                  first_scene_id=self.first_scene_id, n_actions_1=self.n_actions_1)   # :551-556
        return self._end_round(learn, losses, utd, raw, action, reward, skipped, greedy, outcomes)

    def _rotation_dict(self, row):
        from collections import defaultdict
        d = defaultdict(int)
        for rot, cnt in enumerate(self._rot_counts[row].tolist()):
            if cnt:
                d[str(rot)] = int(cnt)
        return d

    greedy_rotations = property(lambda self: self._rotation_dict(0))                 # :460
    greedy_rotations_successes = property(lambda self: self._rotation_dict(1))       # :462
    random_rotations_successes = property(lambda self: self._rotation_dict(2))       # :465

    def save(self, path):
        """The reference trainer's checkpoint (Grasping_Agent_multidiscrete.py:560-575), key for key: what ``load_path`` of this class and of the reference's ``Grasp_Agent`` read."""
        torch.save({"step": self.steps_done, "model_state_dict": self.policy_net.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
                    "epsilon": self.eps_threshold, "greedy_rotations": dict(self.greedy_rotations), "greedy_rotations_successes": dict(self.greedy_rotations_successes),
                    "random_rotations_successes": dict(self.random_rotations_successes)}, path)

    def _end_round(self, learn, losses, utd, raw, action, reward, skipped, greedy, outcomes):
        self.last_loss = losses[-1] if losses else None
        rot, win = (action // self.n_actions_1).long(), reward.long() == 1                                # update_tensorboard's counters (:448-465), this rank's scenes
        self._rot_counts[0] += torch.bincount(rot[greedy], minlength=self.n_actions_2)
        self._rot_counts[1] += torch.bincount(rot[greedy & win], minlength=self.n_actions_2)
        self._rot_counts[2] += torch.bincount(rot[~greedy & win], minlength=self.n_actions_2)
        if self.shared and learn and losses:
            # The replicas take identical steps on identical batches, but GPU kernels are not bit-reproducible across processes (MIOpen's weight-gradient kernels
            # accumulate with atomics): left alone the copies drift apart at rounding level. One broadcast of rank 0's weights, batch-norm buffers and Adam moments
            # per round (88 MB over RCCL, against seconds of physics; flattened: one collective per dtype class, not one per tensor) makes "one agent" exact
            # again; on the CPU it is a copy of equal values.
            ts = [t.data for t in list(self.policy_net.parameters()) + list(self.policy_net.buffers())]
            ts += [v for st in self.optimizer.state.values() for v in st.values() if torch.is_tensor(v)]
            sharding.broadcast_many_from_rank0(ts)
        self.rounds_done += 1
        return {"observation": raw, "action": action, "reward": reward, "skipped": skipped, "greedy": greedy, "loss": self.last_loss if learn else None, "losses": losses,
                "update_to_data": utd, "outcomes": outcomes, "epsilon": self.eps_threshold}
