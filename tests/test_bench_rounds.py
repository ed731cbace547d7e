"""bench.py's stationary IT1 workload driver (It1Rounds): episode flags, device-side re-aiming, state-tensor aliasing."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mujoco_rl_ur5_amd.native import BatchSim  # noqa: E402


def _run(model, sim, dev, n, rounds, rule="aimed"):
    wl = bench.It1Rounds(torch, model, sim, dev, 0, n, n, rule)
    rew = torch.zeros((rounds, n), dtype=torch.int32, device=dev)
    acts = []
    for r in range(rounds):
        seeds = wl.reset_seeds(r)
        assert [int(x != 0) for x in seeds.tolist()] == [int((r + 1 + g) % bench.EP == 0) for g in range(n)]
        act, pixel = wl.launch(r, rew[r])
        sim.sync()
        acts.append(act.clone())
    return rew, acts


def test_rounds_on_the_emulation_build(model_it1, emul_lib):
    n = 4
    sim = BatchSim(model_it1, n, lib_path=emul_lib)
    sim.reset((20 + np.arange(n)).astype(np.uint64), 1, 1000.0)
    st = sim.state_tensor("cpu")
    assert np.array_equal(st[:, :36].numpy(), sim.get_state()["qpos"])              # aliases the engine's records
    rew, acts = _run(model_it1, sim, torch.device("cpu"), n, 5)
    assert rew.float().mean() > 0.5                                               # aimed at boxes that are really there
    assert all(abs(float(a[:, 2].max()) - 0.91) < 1e-12 for a in acts)


@pytest.mark.gpu
def test_rounds_stay_stationary_on_gpu(model_it1):
    n = 256
    sim = BatchSim(model_it1, n, device_id=0)
    sim.reset((20 + np.arange(n)).astype(np.uint64), 1, 1000.0)
    dev = torch.device("cuda", 0)
    st = sim.state_tensor(dev)
    q1 = sim.get_state()["qpos"]
    assert np.array_equal(st[:, :36].cpu().numpy(), q1)
    q2 = q1.copy()
    q2[:, 8] += 0.001
    sim.set_state(qpos=q2)
    assert np.array_equal(st[:, :36].cpu().numpy(), q2), "state_tensor must alias the engine's device records, not copy them"
    sim.set_state(qpos=q1)
    sim.set_stream(torch.cuda.current_stream().cuda_stream)
    rew, _ = _run(model_it1, sim, dev, n, 9)
    rates = rew.float().mean(dim=1).cpu().numpy()
    assert rates.min() > 0.45 and rates.max() < 0.9, rates                         # the oracle's episode average is 0.65-0.67
    assert abs(rates[1:5].mean() - rates[5:9].mean()) < 0.1, rates                 # same mix in every window of EP rounds
    assert sim.counters()["status"].max() == 0
