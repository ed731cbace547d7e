"""Multi-GPU layer: scenes shard embarrassingly over ranks; the only collective is one all_gather of grasp outcomes.

The reference is single-process; its replay push (Grasping_Agent_multidiscrete.py:551-554) is what
the gathered 16-byte records ``{env_id, pixel, rot, reward}`` feed. One process per GPU, ``torch.distributed`` backend
"nccl" (= RCCL over xGMI) on the GPU box, "gloo" in the CPU tests. Seeds are keyed by the GLOBAL scene id, so a scene's
trajectory does not depend on how the batch is sharded.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_total, rank, world_size):
    """Contiguous scene-id range of ``rank`` (4096 -> 512 per GPU at 8 ranks)."""
    if n_total % world_size:
        raise ValueError("n_total must be divisible by world_size")
    n_local = n_total // world_size
    return rank * n_local, (rank + 1) * n_local


def global_seeds(base_seed, n_total, rank, world_size, episode=0):
    lo, hi = shard_range(n_total, rank, world_size)
    return np.uint64(base_seed) + np.arange(lo, hi, dtype=np.uint64) + np.uint64(episode * n_total)


def pack_outcomes(env_ids, pixels, rots, rewards):
    """int32 [n_local, 4] records (16 B each)."""
    return np.stack([np.asarray(env_ids), np.asarray(pixels), np.asarray(rots), np.asarray(rewards)], axis=1).astype(np.int32)


# A process group of ONE rank normally skips every collective (there is nobody to talk to). FORCE_COLLECTIVES (or UR5_FORCE_COLLECTIVES=1 in the
# environment) sends a one-rank job down the same RCCL calls as an N-rank job -- `gpurun` boxes have one GPU, and this is how the collective path
# (RCCL init, all_gather_into_tensor of the int32 records, the broadcasts incl. the host-tensor detour, the replay's batch all-reduce) is executed
# on a real MI355X at all (tests/test_sharding.py::test_rccl_collectives_execute_on_one_gpu, `bench.py --collectives`). Results are unchanged.
import os as _os
FORCE_COLLECTIVES = _os.environ.get("UR5_FORCE_COLLECTIVES", "0") not in ("", "0")


# Accounting of the collectives (round 6; bench.py's dqn sub-results and `--collectives`): with TIME_COLLECTIVES set, every collective below is bracketed by a pair of CUDA events
# on the current stream (no host synchronisation); collective_stats() resolves them. {name: [calls, payload bytes, device ms]}.
TIME_COLLECTIVES = False
_pending, _stats = [], {}


def _timed(name, nbytes, fn, on_cuda):
    import torch
    if not (TIME_COLLECTIVES and on_cuda):
        st = _stats.setdefault(name, [0, 0, 0.0])
        st[0] += 1; st[1] += int(nbytes)
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    _pending.append((name, int(nbytes), e0, e1))
    return out


def collective_stats(reset=False):
    """{name: {"calls", "bytes", "ms"}} of the collectives issued so far (device time between the bracketing events: on one rank that is RCCL's own cost, the wire adds to it)."""
    import torch
    if _pending:
        torch.cuda.synchronize()
        for name, nbytes, e0, e1 in _pending:
            st = _stats.setdefault(name, [0, 0, 0.0])
            st[0] += 1; st[1] += nbytes; st[2] += e0.elapsed_time(e1)
        _pending.clear()
    out = {k: dict(calls=v[0], bytes=v[1], ms=v[2]) for k, v in _stats.items()}
    if reset:
        _stats.clear()
    return out


def collectives_active():
    """True when the calls below really go through torch.distributed: a process group exists and has peers (or FORCE_COLLECTIVES is set)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def gather_outcomes(records, device=None):
    """all_gather of the per-rank outcome records -> int32 [n_total, 4] on every rank, ordered by rank (= by scene id).

    ``records`` is a numpy int32 [n_local, 4] array or a torch tensor. Falls back to a copy when torch.distributed is not
    initialised (single process).
    """
    import torch
    import torch.distributed as dist
    t = records if isinstance(records, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(records, dtype=np.int32))
    if device is not None:
        t = t.to(device)
    if not collectives_active():
        return t.clone()
    if dist.get_backend() != "nccl" and t.is_cuda:   # gloo (CPU tests, several ranks on one GPU): gather through host memory
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t.cpu().contiguous())
        return torch.cat(parts).to(t.device)
    out = torch.empty((dist.get_world_size() * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
    _timed("all_gather_outcomes", t.numel() * t.element_size(), lambda: dist.all_gather_into_tensor(out, t.contiguous()), t.is_cuda)
    return out


def broadcast_from_rank0(t):
    """In-place broadcast of a tensor from rank 0 (no-op in a single process); gloo with CUDA tensors goes through host memory."""
    import torch.distributed as dist
    if not collectives_active():
        return t
    if dist.get_backend() != "nccl" and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, 0)
        t.copy_(h)
    elif dist.get_backend() == "nccl" and not t.is_cuda:     # e.g. Adam's host-side `step` counters: RCCL only moves device memory
        import torch
        d = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.broadcast(d, 0)
        t.copy_(d.cpu())
    else:
        _timed("broadcast", t.numel() * t.element_size(), lambda: dist.broadcast(t, 0), t.is_cuda)
    return t


def broadcast_many_from_rank0(tensors):
    """In-place broadcast of a LIST of tensors from rank 0 as ONE collective per (dtype, host / device) class: the tensors are flattened into one buffer,
    broadcast, and copied back. The agent's per-round weight / batch-norm / Adam-state refresh is ~400 tensors: one 88 MB broadcast instead of hundreds
    of small ones (round-4 advice). Returns the number of collectives issued (0 in a single process)."""
    import torch
    if not collectives_active():
        return 0
    classes = {}
    for t in tensors:
        classes.setdefault((t.dtype, t.is_cuda), []).append(t)
    for (_, _), ts in sorted(classes.items(), key=lambda kv: str(kv[0])):
        flat = torch.cat([t.detach().reshape(-1) for t in ts]) if len(ts) > 1 else ts[0].detach().reshape(-1).clone()
        broadcast_from_rank0(flat)
        o = 0
        for t in ts:
            n = t.numel()
            t.detach().copy_(flat[o:o + n].view(t.shape))
            o += n
    return len(classes)


# ------------------------------------------------------------------ random numbers keyed by GLOBAL scene id
# A generator per rank would hand rank 1's first scene the numbers rank 0's first scene gets: the agent's exploration, depth noise and colour
# jitter would depend on how the batch is sharded. These draws are a pure function of (seed, global scene id, round, stream, index) instead --
# SplitMix64's finaliser over a counter, evaluated with torch integer ops on whatever device holds the scene ids -- so N ranks with n scenes each
# draw exactly what one rank with N n scenes draws (tests/test_sharding.py).
_M1, _M2, _GOLD = 0xBF58476D1CE4E5B9 - (1 << 64), 0x94D049BB133111EB - (1 << 64), 0x9E3779B97F4A7C15 - (1 << 64)


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)          # logical shift of an int64 tensor


def _mix64(z):
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def scene_uniform(seed, gids, round_index, stream, count, dtype=None):
    """U[0, 1) draws [len(gids), count] for the scenes with global ids ``gids`` (int64 tensor) in round ``round_index`` of random stream ``stream``."""
    import torch
    dtype = dtype or torch.float32
    # the scalar part of the key is reduced modulo 2^64 and folded to a signed value first: python ints do not wrap, and an offset beyond 64 bits
    # (any seed above ~600) cannot be added to an int64 tensor
    off = ((int(seed) * 0x632BE5AB + int(round_index)) * 0x1000003 + int(stream)) & ((1 << 64) - 1)
    off -= (off >> 63) << 64
    key = _mix64(gids.to(torch.int64) * _GOLD + off)       # one 64-bit key per scene
    z = _mix64(key[:, None] + torch.arange(1, count + 1, dtype=torch.int64, device=gids.device)[None, :] * _GOLD)
    if dtype == torch.float64:
        return _lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)
    return _lsr(z, 40).to(torch.float32) * (1.0 / 16777216.0)


def scene_normal(seed, gids, round_index, stream, count):
    """N(0, 1) draws [len(gids), count], float32 (Box-Muller on two uniform streams)."""
    import math
    import torch
    u = scene_uniform(seed, gids, round_index, stream, 2 * count)
    return torch.sqrt(-2.0 * torch.log(1.0 - u[:, :count])) * torch.cos((2.0 * math.pi) * u[:, count:])
