"""Batched ``GraspEnv``: the reference's gym environment surface on top of the HIP engine.

Mirrors ``gym_grasper/envs/GraspingEnv.py`` (class ``GraspEnv``, :25): ``step(action, record_grasps, markers,
action_info) -> (obs, reward, done, info)``, ``reset() -> obs``, ``action_space = MultiDiscrete([H*W, 6])``,
``rotations``, ``TABLE_HEIGHT``, ``IMAGE_WIDTH/HEIGHT``, ``current_observation``, ``model``, ``controller``.
With ``n_envs == 1`` shapes equal the reference's; otherwise actions are ``int[N, 2]``, observations and rewards gain a
leading N. The whole 12-phase ``move_and_grasp`` script (:205-386) of every scene runs inside ONE kernel launch.

Observation: ``observation="render"`` (default) is the reference's RGB-D observation -- ``get_image_data`` +
``depth_2_meters`` (GraspingEnv.py:390-406), ray-cast on the GPU (csrc/ur5_raster.h) -- so the grasp height comes from
the depth image (IT4+, GraspingEnv.py:100-104,258-259). ``observation="flat"`` is the IT1 setting of README.md:20 ("fixed
z-coordinate for grasping"): depth = camera height - TABLE_HEIGHT everywhere, rgb zeros, nothing rendered.
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np

from .controller import MJ_Controller
from .model import CompiledModel, load_model
from .native import BatchSim


class MultiDiscrete:
    """Minimal stand-in for ``gym.spaces.MultiDiscrete`` (gym is not a dependency of the engine)."""

    def __init__(self, nvec, seed=None):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return (self._rng.random(self.nvec.shape) * self.nvec).astype(np.int64)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.nvec.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

    def __repr__(self):
        return "MultiDiscrete({})".format(self.nvec.tolist())


class GraspEnv(object):
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 500}

    def __init__(self, file="/UR5+gripper/UR5gripper_2_finger_many_objects.xml", image_width=200, image_height=200, show_obs=True,
                 demo=False, render=False, n_envs=1, device_id=0, observation="render", check_mode=0, base_seed=20, first_scene_id=0, n_total=None,
                 _lib_path=None):
        self.initialized = False
        self.IMAGE_WIDTH = image_width
        self.IMAGE_HEIGHT = image_height
        self.rotations = {0: 0, 1: 30, 2: 60, 3: 90, 4: -30, 5: -60}       # GraspingEnv.py:40
        self.action_space_type = "multidiscrete"
        self.step_called = 0
        self.n_envs = int(n_envs)
        self.model = file if isinstance(file, CompiledModel) else load_model(file)
        self.frame_skip = 1                                                  # MujocoEnv.__init__(full_path, 1), :47
        self.sim = BatchSim(self.model, self.n_envs, device_id=device_id, lib_path=_lib_path)
        self.viewer = None
        self._set_action_space()
        self.controller = MJ_Controller(self.model, self.sim, self.viewer)   # :51
        self.initialized = True
        self.grasp_counter = 0
        self.show_observations = show_obs
        self.demo_mode = demo
        self.TABLE_HEIGHT = 0.91                                             # :56
        self.render = render
        if observation not in ("flat", "render"):
            raise ValueError("observation must be 'flat' or 'render'")
        self.observation_mode = observation
        # 0 = in-tree script, 1 = IT1 (README.md:20), 2 = in-tree script in demo mode: close_gripper(max_steps=100) as the final check (:318-321)
        self.check_mode = 2 if (demo and check_mode == 0) else check_mode
        self.base_seed = base_seed                                           # Grasping_Agent_multidiscrete.py:64
        # multi-rank runs: this handle simulates scenes [first_scene_id, first_scene_id + n_envs) of n_total; seeds are keyed by the
        # GLOBAL scene id (sharding.global_seeds), so a scene's trajectory does not depend on how the batch is sharded
        self.first_scene_id = int(first_scene_id)
        if n_total is None:
            # the episode stride of the seeds must be the GLOBAL scene count on every rank (else rank 0's episode 1 re-simulates rank 1's episode 0)
            world = 1
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    world = dist.get_world_size()
            except ImportError:
                pass
            if world > 1 and self.first_scene_id == dist.get_rank() * self.n_envs:
                n_total = world * self.n_envs                                # this rank's contiguous shard (sharding.shard_range)
            elif self.first_scene_id == 0:
                n_total = self.n_envs                                        # a stand-alone handle (also on rank 0 of a multi-rank job that shards by hand: pass n_total there)
            else:
                raise ValueError("first_scene_id > 0 needs n_total (the global scene count): seeds are keyed by global scene id and episode, "
                                 "and a per-rank stride would make ranks re-simulate each other's episodes")
        self.n_total = int(n_total)
        if self.first_scene_id + self.n_envs > self.n_total:
            raise ValueError("scene range [first_scene_id, first_scene_id + n_envs) exceeds n_total")
        self.device_id = int(device_id)
        self._episode = 0
        self.last_phase_steps = None
        self.last_phase_result = None
        self.current_observation = self.get_observation(show=False)

    def __repr__(self):
        return f"GraspEnv(obs height={self.IMAGE_HEIGHT}, obs_width={self.IMAGE_WIDTH}, AS={self.action_space_type})"

    @property
    def dt(self):
        return self.model.opt["timestep"] * self.frame_skip

    def _one(self, v):
        return v[0] if self.n_envs == 1 else v

    def _set_action_space(self):                                             # :158-167
        self.action_space = MultiDiscrete([self.IMAGE_HEIGHT * self.IMAGE_WIDTH, len(self.rotations)])
        return self.action_space

    # ------------------------------------------------------------------ step / reset
    def step(self, action, record_grasps=False, markers=False, action_info="no info"):   # :62-156
        done = False
        info = {}
        if self.step_called == 1:
            self.current_observation = self.get_observation(show=False)      # :87-88
        a = np.atleast_2d(np.asarray(action, dtype=np.int64))
        if a.shape != (self.n_envs, 2):
            raise ValueError(f"action must have shape ({self.n_envs}, 2) = [pixel index, rotation index]")
        x = a[:, 0] % self.IMAGE_WIDTH                                       # :95
        y = a[:, 0] // self.IMAGE_WIDTH                                      # :96
        rotation = a[:, 1]                                                   # :97
        depth_img = np.asarray(self.current_observation["depth"]).reshape(self.n_envs, self.IMAGE_HEIGHT, self.IMAGE_WIDTH)
        depth = depth_img[np.arange(self.n_envs), y, x]                      # :100
        coords = self.controller.pixel_2_world_batch(x, y, depth, width=self.IMAGE_WIDTH, height=self.IMAGE_HEIGHT)   # :102-104
        skip = (coords[:, 2] < 0.8) | (coords[:, 1] > -0.3)                  # :124
        reward = self.move_and_grasp(coords, rotation, skip=skip)
        self.current_observation = self.get_observation(show=self.show_observations)   # :152
        self.step_called += 1
        info["phase_steps"] = self._one(self.last_phase_steps)
        info["skipped"] = self._one(skip)
        info["status"] = self._one(self.check_status())
        return self.current_observation, self._one(reward), done, info

    # ------------------------------------------------------------------ soft failures of a scene (per-env status for soft failures)
    STATUS_BITS = {1: "contact slots exhausted: contacts were dropped (UR5_ST_CONTACT_OVERFLOW; 30 slots per small scene, 96 per 40-object pile)",
                   2: "a step produced a non-finite state: the scene was reset to qpos0 as mj_step does (UR5_ST_NAN)",
                   4: "constraint-row / envelope lists exhausted (UR5_ST_ROW_OVERFLOW)", 8: "broad-phase candidate list exhausted: pairs were dropped (UR5_ST_CAND_OVERFLOW)"}

    status_check_every = 16   # step_device(sync=True) reads the status words on its 1st, 17th, ... call (round-5 advice: not a blocking read per round for a once-only warning)

    def scene_status(self):
        """int64 [n_envs]: the engine's status bits of every scene, of its running episode and of episodes that ended inside a fused launch (0 = clean)."""
        c = self.sim.counters()
        return c["status"] | c["status_ended"]

    def check_status(self):
        """Reads the status words and warns ONCE per distinct bit when a scene is flagged: its rewards come from a simulation that dropped contacts / rows or was
        reset, and what survives a contact-slot overflow depends on the order the slots were claimed in (round-4 advice). ``self.last_status`` keeps the words."""
        st = self.scene_status()
        self.last_status = st
        seen = getattr(self, "_status_warned", 0)
        new = int(np.bitwise_or.reduce(st)) & ~seen if len(st) else 0
        if new:
            import warnings
            self._status_warned = seen | new
            for bit, what in self.STATUS_BITS.items():
                if new & bit:
                    warnings.warn(f"GraspEnv: {int((st & bit != 0).sum())} of {self.n_envs} scenes flagged -- {what}; mask their rewards with env.last_status", RuntimeWarning)
        return st

    def move_and_grasp(self, coordinates, rotation, render=False, record_grasps=False, markers=False, plot=False, skip=None):
        """GraspingEnv.py:205-386 for every scene at once; scenes flagged in ``skip`` (:124-131) sit the launch out."""
        coords = np.atleast_2d(np.asarray(coordinates, dtype=np.float64))
        rot = np.broadcast_to(np.asarray(rotation, dtype=np.int64), (self.n_envs,))
        # a skipped scene takes the kernel's early-out (script case 0), exactly as in step_device: nothing of its state changes
        rew, ps, pr = self.sim.grasp_attempt(coords, rot, check_mode=self.check_mode, table_height=self.TABLE_HEIGHT, skip=skip)
        self.last_phase_steps, self.last_phase_result = ps, pr
        self.controller.last_movement_steps = self._one(ps[:, 11])
        return rew.astype(np.int64)

    def episode_seeds(self, episode):
        """Seeds of this handle's scenes for ``episode``: base + global scene id + episode * n_total (== sharding.global_seeds for the rank that
        owns this scene range), so a scene's trajectory does not depend on how the batch is sharded."""
        return (np.uint64(self.base_seed) + np.arange(self.first_scene_id, self.first_scene_id + self.n_envs, dtype=np.uint64)
                + np.uint64(episode * self.n_total))

    def reset(self):
        """MujocoEnv.reset() [3P] -> reset_model() (GraspingEnv.py:409-477)."""
        return self.reset_model()

    def reset_model(self, show_obs=True):                                    # :409-477
        seeds = self.episode_seeds(self._episode)
        self._episode += 1
        self.sim.reset(seeds, mode=1, settle_ms=1000.0 + (5000.0 if self.demo_mode else 0.0))   # :473-475
        self.current_observation = self.get_observation(show=self.show_observations)
        return self.current_observation

    def get_observation(self, show=True):                                    # :390-406
        if self.observation_mode == "render":
            rgb, depth = self.sim.render(self.model.camera_name2id("top_down"), self.IMAGE_WIDTH, self.IMAGE_HEIGHT, depth_mode=1)
            depth = self.controller.depth_2_meters(depth).astype(np.float32)  # :398-399
        else:
            cam_z = self.model.cam_pos0[self.model.camera_name2id("top_down")][2]
            depth = np.full((self.n_envs, self.IMAGE_HEIGHT, self.IMAGE_WIDTH), cam_z - 0.91, dtype=np.float32)
            rgb = np.zeros((self.n_envs, self.IMAGE_HEIGHT, self.IMAGE_WIDTH, 3), dtype=np.uint8)
        observation = defaultdict()
        observation["rgb"] = self._one(rgb)
        observation["depth"] = self._one(depth)
        return observation

    # ------------------------------------------------------------------ device-resident variants
    # The agent loop of Grasping_Agent_multidiscrete.py needs, per step, the RGB-D observation as a network input and the depth under
    # the chosen pixel. With thousands of scenes the 280 KB observation per scene must not cross PCIe: these two methods keep
    # observations, actions and rewards in torch tensors on the simulating GPU (host tensors for the CPU lane-emulation build).
    def _torch_device(self, device):
        """None -> the GPU this handle simulates on. The engine dereferences raw data_ptr()s, so a CUDA tensor on another device is an error."""
        import torch
        d = torch.device(f"cuda:{self.device_id}" if device is None else device)
        if d.type == "cuda" and d.index is None:
            d = torch.device("cuda", torch.cuda.current_device())
        if d.type == "cuda" and d.index != self.device_id:
            raise ValueError(f"tensors on {d} but the engine handle runs on cuda:{self.device_id}")
        return d

    def _torch_setup(self, device):
        import torch
        device = self._torch_device(device)
        if getattr(self, "_tdev", None) == str(device):
            return torch
        self._tdev = str(device)
        n, H, W = self.n_envs, self.IMAGE_HEIGHT, self.IMAGE_WIDTH
        self._t_rgb = torch.zeros((n, H, W, 3), dtype=torch.uint8, device=device)
        self._t_gl = torch.zeros((n, H, W), dtype=torch.float32, device=device)
        self._t_act = torch.zeros((n, 8), dtype=torch.float64, device=device)
        self._t_rew = torch.zeros((n,), dtype=torch.int32, device=device)
        self.controller.create_camera_data(W, H, "top_down")
        Kinv, Rinv = np.linalg.inv(self.controller.cam_matrix), np.linalg.inv(self.controller.cam_rot_mat)
        ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        rays = (Rinv @ Kinv @ np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])).T           # pos_w = t - depth * ray (pixel_2_world, :783-806)
        self._t_rays = torch.from_numpy(rays.reshape(H, W, 3)).to(device)
        self._t_t = torch.from_numpy(Rinv @ self.controller.cam_pos).to(device)
        return torch

    def use_stream(self, stream):
        """Run this env's engine launches on a torch CUDA stream (ur5_set_stream): kernels are then ordered with the torch work queued on that stream, and the
        ``sync=False`` forms of observation_device / step_device need no host synchronisation at all -- the caller works under ``torch.cuda.stream(stream)`` and
        waits for the stream when it wants the results (agent.BatchedGraspAgent with pipeline_groups > 1: one env + stream per scene group)."""
        self.sim.set_stream(stream.cuda_stream)
        self._stream = stream

    def observation_device(self, device=None, sync=True):
        """get_observation (:390-406) without leaving the device: {"rgb": uint8 [N,H,W,3], "depth": float32 [N,H,W] metres}."""
        torch = self._torch_setup(device)
        cam = self.model.camera_name2id("top_down")
        if self._t_rgb.is_cuda and sync:
            torch.cuda.synchronize()
        self.sim.render_dev(self._t_rgb.data_ptr(), self._t_gl.data_ptr(), cam, self.IMAGE_WIDTH, self.IMAGE_HEIGHT, 1)
        if sync:
            self.sim.sync()
        ext = self.model.opt["extent"]
        near, far = self.model.opt["znear"] * ext, self.model.opt["zfar"] * ext
        return {"rgb": self._t_rgb, "depth": near / (1 - self._t_gl * (1 - near / far))}          # depth_2_meters (:729-740)

    def pixel_world_device(self, depth, device=None):
        """World coordinates [N,H,W,3] (float64) of every pixel given the metric depth image: pixel_2_world for all pixels at once."""
        self._torch_setup(device)
        return self._t_t - depth.double().unsqueeze(-1) * self._t_rays

    def step_device(self, action, depth, device=None, sync=True):
        """GraspEnv.step (:62-156) with ``action`` long [N,2] = [pixel, rotation] and the current metric ``depth`` [N,H,W] on the device.
        Returns (reward int32 [N], skipped bool [N]) as device tensors; the caller asks for the next observation when it needs one.
        sync=False (after use_stream): the launch is only queued; the returned reward tensor is the env's own buffer, valid once the stream has run."""
        torch = self._torch_setup(device)
        a = action.to(self._t_act.device).long().reshape(self.n_envs, 2)
        x, y = a[:, 0] % self.IMAGE_WIDTH, a[:, 0] // self.IMAGE_WIDTH                           # :95-96
        e = torch.arange(self.n_envs, device=a.device)
        d = depth[e, y, x].double()                                                              # :100
        coords = self._t_t - d.unsqueeze(-1) * self._t_rays[y, x]                                # :102-104
        skip = (coords[:, 2] < 0.8) | (coords[:, 1] > -0.3)                                      # :124
        self._t_act.zero_()
        self._t_act[:, :3] = coords
        self._t_act[:, 3] = a[:, 1].double()
        self._t_act[:, 4] = skip.double()
        if self._t_act.is_cuda and sync:
            torch.cuda.synchronize()
        self.sim.grasp_attempt_dev(self._t_act.data_ptr(), self._t_rew.data_ptr(), check_mode=self.check_mode, table_height=self.TABLE_HEIGHT)
        self.step_called += 1
        if not sync:
            return self._t_rew, skip
        self.sim.sync()
        if self.step_called % self.status_check_every == 1 or self.status_check_every <= 1:      # the status words are a device-to-host read of every scene's counters: sampled
            self.check_status()                                                                  # (the bits are sticky until the scene's reset; check_status() reads them on request)
        return self._t_rew.clone(), skip

    def close(self):
        self.sim.close()

    def print_info(self):                                                    # :483-489
        print("Model timestep:", self.model.opt["timestep"])
        print("Set number of frames skipped: ", self.frame_skip)
        print("dt = timestep * frame_skip: ", self.dt)
        print("Frames per second = 1/dt: ", self.metadata["video.frames_per_second"])
        print("Actionspace: ", self.action_space)


def make(id="gym_grasper:Grasper-v0", **kwargs):
    """``gym.make("gym_grasper:Grasper-v0", ...)`` without gym (gym_grasper/__init__.py:4-7)."""
    if id not in ("gym_grasper:Grasper-v0", "Grasper-v0"):
        raise ValueError(f"unknown environment id {id!r}")
    return GraspEnv(**kwargs)
