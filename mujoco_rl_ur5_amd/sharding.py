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
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t.clone()
    if dist.get_backend() != "nccl" and t.is_cuda:   # gloo (CPU tests, several ranks on one GPU): gather through host memory
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t.cpu().contiguous())
        return torch.cat(parts).to(t.device)
    out = torch.empty((dist.get_world_size() * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out
