"""Offline-RL data files in the reference's layout (Offline RL/generate_data.py, grasping_dataset.py)."""
import numpy as np
import torch

from mujoco_rl_ur5_amd.dataset import FILE_SIZE, GraspingDataWriter, Grasping_Dataset


def test_files_have_the_reference_layout(tmp_path):
    w = GraspingDataWriter(str(tmp_path / "Data"))
    rng = np.random.default_rng(0)
    for r in range(5):                                                       # 5 rounds of 5 scenes -> 2 full files + 1 pending
        obs = {"rgb": rng.integers(0, 255, (5, 8, 8, 3), dtype=np.uint8), "depth": (0.9 + 0.3 * rng.random((5, 8, 8))).astype(np.float32)}
        w.add(obs, torch.tensor([[3, 1], [7, 0], [63, 5], [0, 0], [9, 2]]), np.array([0, 1, 0, 0, 1]))
    assert [f.split("/")[-1] for f in w.files] == ["grasping_data_1.pt", "grasping_data_2.pt"]
    w.flush()
    assert w.files[-1].endswith("grasping_data_3.pt")
    d = torch.load(w.files[0], weights_only=False)                          # exactly what generate_data.py:69-84 saves
    assert sorted(d.keys()) == ["actions", "rewards", "states"] and len(d["states"]) == len(d["actions"]) == len(d["rewards"]) == FILE_SIZE
    assert d["states"][0]["rgb"].shape == (8, 8, 3) and d["states"][0]["rgb"].dtype == np.uint8 and d["states"][0]["depth"].dtype == np.float32
    assert d["actions"][:5] == [1 * 64 + 3, 7, 5 * 64 + 63, 0, 2 * 64 + 9] and d["rewards"][:5] == [0, 1, 0, 0, 1]
    ds = Grasping_Dataset(w.files[0], seed=1)
    x, a, r = ds[2]
    assert len(ds) == FILE_SIZE and x.shape == (4, 8, 8) and x.dtype == torch.float32 and a == 5 * 64 + 63 and r == 0
    assert float(x[3].min()) == 0.0 and float(x[3].max()) == 1.0 and float(x[:3].max()) <= 1.0
    clean = ds.transform_observation(ds.state_list[2], jitter_and_noise=False)[3].numpy()
    dep = np.minimum(d["states"][2]["depth"].astype(np.float64), 1.1) * -1
    assert np.allclose(clean, (dep - dep.min()) / (dep.max() - dep.min()), atol=1e-6)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=4)))       # train.py:73
    assert batch[0].shape == (4, 4, 8, 8) and batch[1].shape == (4,)


def test_a_written_file_read_by_the_references_own_class():
    """tests/golden/offline_rl/grasping_data_1.pt was written by GraspingDataWriter and read by the reference's Offline RL/grasping_dataset.py
    (tools/gen_golden_dataset.py): the reference accepted the layout, and this package's reader returns the same items from the same file."""
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = json.load(open(os.path.join(gold, "dataset_reference.json")))
    ds = Grasping_Dataset(os.path.join(gold, g["file"]), seed=0)
    assert len(ds) == g["len"] == FILE_SIZE
    items = [ds[i] for i in range(len(ds))]
    assert [int(a) for _, a, _ in items] == g["actions"] and [int(r) for _, _, r in items] == g["rewards"]
    assert list(items[0][0].shape) == g["shape"] and str(items[0][0].dtype) == g["dtype"]
    for i, ref in enumerate(g["states"]):
        ref = np.array(ref).reshape(g["shape"])
        mine = ds.transform_observation(ds.state_list[i], jitter_and_noise=False).numpy()
        assert np.array_equal(mine[:3], ref[:3].astype(np.float32))                 # ToTensor: rgb / 255, channel first
        assert np.abs(mine[3] - ref[3]).max() < 0.03                                 # depth: the reference's draw of its N(0, 1e-3) noise apart


def _generate_and_check(tmp_path, device, lib, width, n_envs):
    """generate_data.py's loop on the batched agent: every transition of every round lands in the files, in scene order, with the raw observation the
    action was chosen in; the files load through Grasping_Dataset (whose items the reference's own class reproduces, test above)."""
    from mujoco_rl_ur5_amd.agent import BatchedGraspAgent
    from mujoco_rl_ur5_amd.envs import GraspEnv
    from mujoco_rl_ur5_amd.generate_data import generate_data
    from mujoco_rl_ur5_amd.model import load_model
    env = GraspEnv(file=load_model("it1_4box"), n_envs=n_envs, show_obs=False, observation="render", image_width=width, image_height=width, check_mode=1,
                   _lib_path=lib)
    agent = BatchedGraspAgent(env=env, device=device, mem_size=100, seed=122)
    seen = []
    orig = agent.round

    def spy(**kw):
        out = orig(**kw)
        seen.append((out["observation"]["depth"].cpu().numpy().copy(), out["action"].cpu().numpy().copy(), out["reward"].cpu().numpy().copy()))
        return out
    agent.round = spy
    files, counter = generate_data(agent, str(tmp_path / "Data"), episodes=2, steps=3, learn=True)
    total = 2 * 3 * n_envs
    assert len(files) == -(-total // FILE_SIZE) and sum(counter.values()) == total and env._episode >= 2
    acts = np.concatenate([a for _, a, _ in seen])
    rews = np.concatenate([r for _, _, r in seen])
    deps = np.concatenate([d for d, _, _ in seen])
    got_a, got_r, k = [], [], 0
    for f in files:
        ds = Grasping_Dataset(f, seed=0)
        for i in range(len(ds)):
            x, a, r = ds[i]
            assert x.shape == (4, width, width) and np.array_equal(ds.state_list[i]["depth"], deps[k])       # the observation of THAT transition
            got_a.append(int(a)); got_r.append(int(r)); k += 1
    assert got_a == acts.tolist() and got_r == rews.tolist() and k == total
    assert max(got_a) < 6 * width * width and set(got_r) <= {0, 1}
    env.close()


def test_generate_data_loop_fills_the_files_from_the_batched_agent(tmp_path, emul_lib):
    _generate_and_check(tmp_path, "cpu", emul_lib, 24, 3)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_generate_data_on_gpu(tmp_path):
    _generate_and_check(tmp_path, "cuda", None, 200, 8)
