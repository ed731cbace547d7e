#!/usr/bin/env python3
"""Round trip of an offline-RL data file through the REFERENCE's own reader (Offline RL/grasping_dataset.py): a file written by
mujoco_rl_ur5_amd.dataset.GraspingDataWriter is loaded by the reference's Grasping_Dataset class and the items it returns are stored as
golden vectors (tests/golden/dataset_reference.json + the file itself). Runs in the build container only (needs /root/reference).

torchvision is not installed here: the three transforms the class composes are stubbed with what they do to the data format --
ToPILImage / ColorJitter as identities (the colour jitter is random and tested separately), ToTensor as HWC uint8 -> CHW float / 255."""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, ROOT)
tv, T = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")


class _Compose:
    def __init__(self, ts): self.ts = ts
    def __call__(self, x):
        for t in self.ts: x = t(x)
        return x


T.Compose = _Compose
T.ToPILImage = lambda: (lambda x: x)
T.ColorJitter = lambda **kw: (lambda x: x)
T.ToTensor = lambda: (lambda x: torch.from_numpy(np.ascontiguousarray(x)).permute(2, 0, 1).float() / 255.0)
tv.transforms = T
sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, T
pt = types.ModuleType("prettytable"); pt.PrettyTable = object
sys.modules["prettytable"] = pt
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "Offline RL"))
import grasping_dataset as RD  # noqa: E402
from mujoco_rl_ur5_amd.dataset import GraspingDataWriter  # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
w = GraspingDataWriter(os.path.join(gold, "offline_rl"), file_size=12)
rng = np.random.default_rng(11)
obs = {"rgb": rng.integers(0, 255, (12, 6, 6, 3), dtype=np.uint8), "depth": (0.85 + 0.4 * rng.random((12, 6, 6))).astype(np.float32)}
pix_rot = np.stack([rng.integers(0, 36, 12), rng.integers(0, 6, 12)], axis=1)
w.add(obs, pix_rot, rng.integers(0, 2, 12))
path = w.files[0]
_load = torch.load
torch.load = lambda f, **kw: _load(f, **{"weights_only": False, **kw})      # the reference predates torch 2.6's weights_only=True default
ds = RD.Grasping_Dataset(path)                                   # the reference's class reads the file this package wrote
torch.load = _load
np.random.seed(0)
items = [ds[i] for i in range(len(ds))]
json.dump({"file": os.path.relpath(path, gold), "len": len(ds), "actions": [int(a) for _, a, _ in items], "rewards": [int(r) for _, _, r in items],
           "shape": list(items[0][0].shape), "dtype": str(items[0][0].dtype),
           "states": [[float(v) for v in x.reshape(-1)] for x, _, _ in items[:3]]},
          open(os.path.join(gold, "dataset_reference.json"), "w"))
print("wrote", path, len(ds), items[0][0].shape)
