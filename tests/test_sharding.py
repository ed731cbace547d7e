"""N>1 path on CPU: two gloo ranks shard the scenes, gather the 16-byte outcome records, and reproduce the 1-rank result."""
import os
import sys

import numpy as np
import pytest

from mujoco_rl_ur5_amd import sharding


def test_shard_ranges_and_seeds():
    assert sharding.shard_range(4096, 3, 8) == (1536, 2048)
    with pytest.raises(ValueError):
        sharding.shard_range(10, 0, 4)
    s = np.concatenate([sharding.global_seeds(20, 16, r, 4) for r in range(4)])
    assert np.array_equal(s, 20 + np.arange(16, dtype=np.uint64))


def _worker(rank, world, port, emul_lib, out):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mujoco_rl_ur5_amd.model import load_model
    from mujoco_rl_ur5_amd.native import BatchSim
    from conftest import aimed_actions
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 4
    lo, hi = sharding.shard_range(n_total, rank, world)
    sim = BatchSim(load_model("it1_4box"), hi - lo, lib_path=emul_lib)
    sim.reset(sharding.global_seeds(20, n_total, rank, world), 1, 300.0)
    acts = aimed_actions(sim.get_state()["qpos"], 4, first_id=lo)
    rew, ps, pr = sim.grasp_attempt(acts, rot=0, check_mode=0)
    rec = sharding.pack_outcomes(np.arange(lo, hi), np.arange(lo, hi) * 7, np.zeros(hi - lo), rew)
    allrec = sharding.gather_outcomes(rec).numpy()
    dist.barrier()
    if rank == 0:
        np.save(out, np.concatenate([allrec.ravel(), ps.ravel()]))
    dist.destroy_process_group()


def test_two_rank_gloo_gather_equals_single_process(emul_lib, model_it1, tmp_path):
    import torch.multiprocessing as mp
    from mujoco_rl_ur5_amd.native import BatchSim
    from conftest import aimed_actions
    out = str(tmp_path / "gathered.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, emul_lib, out), nprocs=2, join=True)
    got = np.load(out)
    rec = got[:16].reshape(4, 4)
    sim = BatchSim(model_it1, 4, lib_path=emul_lib)
    sim.reset(20 + np.arange(4, dtype=np.uint64), 1, 300.0)
    rew, ps, pr = sim.grasp_attempt(aimed_actions(sim.get_state()["qpos"], 4), rot=0, check_mode=0)
    assert rec[:, 0].tolist() == [0, 1, 2, 3] and rec[:, 3].tolist() == rew.tolist()
    assert got[16:].astype(int).tolist() == ps[:2].ravel().tolist()       # rank 0's scenes: same step counts as the 1-rank run
