"""The reference's offline-RL data files, written from the batched simulator and read back.

Format (``Offline RL/generate_data.py:21,69-90``): ``Data/grasping_data_{k}.pt`` = ``torch.save({"states": [...], "actions": [...],
"rewards": [...]})`` with FILE_SIZE = 12 entries per file; a state is the raw observation dict the action was chosen in
(``{"rgb": uint8 [H,W,3], "depth": float32 [H,W]}``, ``GraspingEnv.py:390-406``), an action the flat index into the [6, H, W] Q maps
(``transform_action``, ``Grasping_Agent_multidiscrete.py:380-385``), a reward 0/1. ``Grasping_Dataset`` mirrors ``Offline
RL/grasping_dataset.py:12-71`` (depth clipped at 1.1 m, noise, negation, per-image min-max; rgb / 255 -- the torchvision colour
jitter is not applied) so the reference's ``train.py`` loop runs on files written here and vice versa.
"""
from __future__ import annotations

import os

import numpy as np
import torch

FILE_SIZE = 12                                                              # generate_data.py:21


class GraspingDataWriter:
    """Collects (state, action, reward) triples -- batched or one at a time -- and writes a file every ``file_size`` triples."""

    def __init__(self, directory="Data", file_size=FILE_SIZE, first_index=1):
        self.directory, self.file_size, self.number_saved = directory, int(file_size), first_index - 1
        self._s, self._a, self._r = [], [], []
        os.makedirs(directory, exist_ok=True)
        self.files = []

    def add(self, observation, action, reward):
        """observation: dict of arrays / tensors with or without a leading scene axis; action: flat index (or [pixel, rot] pairs,
        converted with n_pixels = H*W); reward: 0/1."""
        rgb = _np(observation["rgb"])
        depth = _np(observation["depth"])
        if rgb.ndim == 3:
            rgb, depth = rgb[None], depth[None]
        act = np.atleast_1d(_np(action))
        if act.ndim == 2:                                                   # [pixel, rot] -> flat (rot-major, = view(-1) of [6,H,W])
            act = act[:, 1] * (rgb.shape[1] * rgb.shape[2]) + act[:, 0]
        rew = np.atleast_1d(_np(reward))
        for e in range(rgb.shape[0]):
            self._s.append({"rgb": np.ascontiguousarray(rgb[e], dtype=np.uint8), "depth": np.ascontiguousarray(depth[e], dtype=np.float32)})
            self._a.append(int(act[e]))
            self._r.append(int(rew[e]))
            if len(self._s) == self.file_size:
                self.flush()

    def flush(self):
        if not self._s:
            return None
        self.number_saved += 1
        name = os.path.join(self.directory, f"grasping_data_{self.number_saved}.pt")      # generate_data.py:83
        torch.save({"states": self._s, "actions": self._a, "rewards": self._r}, name)
        self.files.append(name)
        self._s, self._a, self._r = [], [], []
        return name


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class Grasping_Dataset(torch.utils.data.Dataset):
    """``Offline RL/grasping_dataset.py:12-71``: items are ``[state float32 [4,H,W], action int, reward int]``."""

    def __init__(self, file, seed=None):
        data = torch.load(file, weights_only=False)
        self.state_list, self.action_list, self.reward_list = data["states"], data["actions"], data["rewards"]
        self._rng = np.random.default_rng(seed)

    def __len__(self):
        return len(self.state_list)

    def __getitem__(self, idx):
        return [self.transform_observation(self.state_list[idx]), self.action_list[idx], self.reward_list[idx]]

    def transform_observation(self, observation, normalize=True, jitter_and_noise=True):
        depth = np.array(observation["depth"], dtype=np.float64)
        depth[depth > 1.1] = 1.1                                            # :43-44
        if jitter_and_noise:
            depth += self._rng.normal(loc=0, scale=0.001, size=depth.shape)  # :50
        depth *= -1
        depth = (depth - depth.min()) / (depth.max() - depth.min())         # :51-54
        rgb = torch.from_numpy(np.ascontiguousarray(observation["rgb"])).permute(2, 0, 1).float() / 255.0
        return torch.cat((rgb, torch.from_numpy(depth[None]).float()), dim=0)
