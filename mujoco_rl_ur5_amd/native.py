"""ctypes binding of the C ABI in include/ur5sim.h (libur5sim.so, built from csrc/ by __graft_entry__.build()).

The loader opens ONLY ``mujoco_rl_ur5_amd/csrc/libur5sim.so`` and raises when it is missing or when no GPU answers:
there is no CPU fallback in the product path. (Tests may hand an explicit library path to :class:`BatchSim` to run the
lane-emulation build of the same engine source; nothing in this package does.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libur5sim.so")

RES_NONE, RES_SUCCESS, RES_MAX_STEPS, RES_IK_FAIL = -1, 0, 1, 2
# limits of the two engine variants (csrc/ur5_devmodel.h): (max objects, debug stride, record stride, max contacts)
_VARIANT = {0: (6, 2048, 192, 30), 1: (40, 4096, 832, 96)}
EXPORTS = ["ur5_last_error", "ur5_create", "ur5_destroy", "ur5_num_envs", "ur5_nq", "ur5_nv", "ur5_nu", "ur5_reset", "ur5_reset_dev", "ur5_kernel_ms_total",
           "ur5_set_state", "ur5_get_state", "ur5_set_ctrl", "ur5_get_ctrl", "ur5_step", "ur5_move_group", "ur5_stay",
           "ur5_move_ee", "ur5_ik", "ur5_grasp_attempt", "ur5_grasp_attempt_dev", "ur5_grasp_attempt_reset_dev", "ur5_grasp_rounds_dev", "ur5_sync", "ur5_set_stream", "ur5_set_order_dev", "ur5_set_order_view_dev", "ur5_set_observation_dev", "ur5_last_launch_ms",
           "ur5_get_counters", "ur5_body_xpos", "ur5_render", "ur5_render_dev", "ur5_state_device_ptr"]
TEST_EXPORTS = ["ur5_forward_debug", "ur5_set_step_cap_dev", "ur5_model_uploads"]   # include/ur5sim_test.h: introspection for tests/ and tools/, not part of the boundary


class Config(C.Structure):
    _fields_ = [("ee_body", C.c_int), ("contacts_enabled", C.c_int), ("pid_dt", C.c_double)]


class AimRule(C.Structure):
    """ur5_aim_rule of include/ur5sim.h: the scripted aiming rule a multi-round launch evaluates in the kernel (ur5_grasp_rounds_dev)."""
    _fields_ = [("kind", C.c_int), ("episode_rounds", C.c_int), ("first_scene_id", C.c_int64), ("n_total", C.c_int64), ("base_seed", C.c_uint64),
                ("plate_half_x", C.c_double), ("plate_centre_y", C.c_double), ("plate_half_y", C.c_double), ("z_min", C.c_double), ("z_max", C.c_double),
                ("grasp_z", C.c_double), ("fallback_x", C.c_double), ("fallback_y", C.c_double),
                ("z_from_depth", C.c_int), ("pad", C.c_int), ("cam_x0", C.c_double), ("cam_y0", C.c_double), ("cam_dx", C.c_double), ("cam_dy", C.c_double), ("cam_z", C.c_double)]


_libs = {}


def load(path=None):
    # UR5SIM_LIB: A/B builds of the SAME HIP engine (tools/, kernel experiments); never a CPU build -- ur5_create still needs the GPU
    path = path or os.environ.get("UR5SIM_LIB") or DEFAULT_LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                           "g.build()'). There is no CPU fallback.")
    L = C.CDLL(path)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    L.ur5_last_error.restype = C.c_char_p
    L.ur5_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(Config), C.POINTER(vp)]
    L.ur5_destroy.argtypes = [vp]
    L.ur5_destroy.restype = None
    for f in ("ur5_num_envs", "ur5_nq", "ur5_nv", "ur5_nu", "ur5_sync"):
        getattr(L, f).argtypes = [vp]
    L.ur5_reset.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int, C.c_double]
    L.ur5_set_stream.argtypes = [vp, vp, C.c_int]
    L.ur5_set_order_dev.argtypes = [vp, vp]
    L.ur5_set_order_view_dev.argtypes = [vp, vp]
    L.ur5_set_observation_dev.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
    L.ur5_reset_dev.argtypes = [vp, vp, vp, C.c_double]
    L.ur5_kernel_ms_total.argtypes = [vp]
    L.ur5_kernel_ms_total.restype = C.c_double
    L.ur5_set_state.argtypes = [vp, dp, dp, dp, dp]
    L.ur5_get_state.argtypes = [vp, dp, dp, dp, dp]
    L.ur5_set_ctrl.argtypes = [vp, dp]
    L.ur5_get_ctrl.argtypes = [vp, dp]
    L.ur5_step.argtypes = [vp, C.c_int]
    L.ur5_move_group.argtypes = [vp, C.POINTER(C.c_uint32), dp, dp, ip, ip, ip]
    L.ur5_stay.argtypes = [vp, C.c_double]
    L.ur5_move_ee.argtypes = [vp, dp, dp, ip, ip, ip]
    L.ur5_ik.argtypes = [vp, dp, dp, ip]
    L.ur5_grasp_attempt.argtypes = [vp, dp, C.POINTER(C.c_uint8), C.c_int, C.c_double, ip, ip, ip]
    L.ur5_grasp_attempt_dev.argtypes = [vp, vp, C.c_int, C.c_double, vp]
    L.ur5_grasp_attempt_reset_dev.argtypes = [vp, vp, C.c_int, C.c_double, vp, vp, C.c_double]
    if hasattr(L, "ur5_grasp_rounds_dev"):
        L.ur5_grasp_rounds_dev.argtypes = [vp, C.POINTER(AimRule), C.c_int, C.c_int, C.c_int, C.c_double, vp, vp, C.c_double]
    L.ur5_last_launch_ms.argtypes = [vp]
    L.ur5_last_launch_ms.restype = C.c_double
    L.ur5_get_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    L.ur5_body_xpos.argtypes = [vp, dp]
    L.ur5_state_device_ptr.argtypes = [vp]
    L.ur5_state_device_ptr.restype = vp
    L.ur5_forward_debug.argtypes = [vp, dp]
    if hasattr(L, "ur5_set_step_cap_dev"):
        L.ur5_set_step_cap_dev.argtypes = [vp, vp]
    if hasattr(L, "ur5_model_uploads"):
        L.ur5_model_uploads.argtypes = [vp]
        L.ur5_model_uploads.restype = C.c_long
    L.ur5_render.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_float)]
    L.ur5_render_dev.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    _libs[path] = L
    return L


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


class BatchSim:
    """N independent scenes on one GPU. Thin, array-in/array-out mirror of the C ABI."""

    def __init__(self, model, n_env, device_id=0, contacts_enabled=True, pid_dt=0.0, lib_path=None):
        self.lib = load(lib_path)
        self.model = model
        self.n = int(n_env)
        blob = model.to_blob()
        cfg = Config(model.body_name2id("ee_link"), int(bool(contacts_enabled)), float(pid_dt))
        h = C.c_void_p()
        rc = self.lib.ur5_create(blob, len(blob), self.n, int(device_id), C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"ur5_create failed ({rc}): {self.lib.ur5_last_error().decode()}")
        self._h = h
        self.nq, self.nv, self.nu = model.nq, model.nv, model.nu
        # which engine of libur5sim.so serves this model (ur5host::count_objects): more than 6 objects, or condim 6 on a collidable geom
        self.variant = 1 if ((model.nv - 8) // 6 > _VARIANT[0][0] or bool(np.any((np.asarray(model.geom_condim) > 4) & (np.asarray(model.geom_collide) != 0)))) else 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.ur5_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.ur5_last_error().decode()}")

    def _full(self, v, dtype):
        a = np.asarray(v, dtype=dtype)
        return np.ascontiguousarray(np.broadcast_to(a, (self.n,) + a.shape[1:] if a.ndim else (self.n,)).copy()) if a.ndim == 0 or a.shape[0] != self.n \
            else np.ascontiguousarray(a)

    # ---- state
    def reset(self, seeds, mode=1, settle_ms=1000.0):
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (self.n,)).copy())
        self._check(self.lib.ur5_reset(self._h, s.ctypes.data_as(C.POINTER(C.c_uint64)), mode, float(settle_ms)), "ur5_reset")

    def reset_dev(self, seeds_ptr, mask_ptr=None, settle_ms=1000.0):
        """Asynchronous reset of the flagged scenes: seeds_ptr -> [n] uint64, mask_ptr -> [n] uint8 or None, HIP device pointers."""
        self._check(self.lib.ur5_reset_dev(self._h, C.c_void_p(seeds_ptr), C.c_void_p(mask_ptr) if mask_ptr else None, float(settle_ms)),
                    "ur5_reset_dev")

    def get_state(self):
        qpos, qvel, warm = np.zeros((self.n, self.nq)), np.zeros((self.n, self.nv)), np.zeros((self.n, self.nv))
        pid = np.zeros((self.n, self.nu, 4))
        self._check(self.lib.ur5_get_state(self._h, _dp(qpos), _dp(qvel), _dp(warm), _dp(pid)), "ur5_get_state")
        return dict(qpos=qpos, qvel=qvel, warmstart=warm, pid=pid)

    def set_state(self, qpos=None, qvel=None, warmstart=None, pid=None):
        def prep(a, shape):
            if a is None:
                return None
            a = np.asarray(a, dtype=np.float64)
            return np.ascontiguousarray(np.broadcast_to(a, shape).copy())
        arrs = [prep(qpos, (self.n, self.nq)), prep(qvel, (self.n, self.nv)), prep(warmstart, (self.n, self.nv)),
                prep(pid, (self.n, self.nu, 4))]
        self._check(self.lib.ur5_set_state(self._h, *[_dp(a) for a in arrs]), "ur5_set_state")

    def set_ctrl(self, ctrl):
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(ctrl, dtype=np.float64), (self.n, self.nu)).copy())
        self._check(self.lib.ur5_set_ctrl(self._h, _dp(c)), "ur5_set_ctrl")

    def get_ctrl(self):
        c = np.zeros((self.n, self.nu))
        self._check(self.lib.ur5_get_ctrl(self._h, _dp(c)), "ur5_get_ctrl")
        return c

    def counters(self):
        c = np.zeros((self.n, 6), dtype=np.int64)
        self._check(self.lib.ur5_get_counters(self._h, c.ctypes.data_as(C.POINTER(C.c_int64))), "ur5_get_counters")
        # column 5 is variant-specific (include/ur5sim.h): reused Newton factors (many-object engine) / steps served by the broad phase's pair cache
        # status: bits of the running episode; status_ended: bits raised in episodes that a fused attempt + reset launch has ended since the last host-side reset
        return dict(total_steps=c[:, 0], last_steps=c[:, 1], status=c[:, 2] & 0xFF, status_ended=(c[:, 2] >> 8) & 0xFF, solver_iters=c[:, 3], ncon_max=c[:, 4],
                    factor_reuse=c[:, 5], cached_broadphase_steps=c[:, 5])

    # ---- dynamics
    def step(self, nsteps=1):
        self._check(self.lib.ur5_step(self._h, int(nsteps)), "ur5_step")

    def move_group(self, mask, target, tol, max_steps):
        mask = np.ascontiguousarray(np.broadcast_to(np.asarray(mask, dtype=np.uint32), (self.n,)).copy())
        tgt = None
        if target is not None:
            t = np.asarray(target, dtype=np.float64)
            t = np.broadcast_to(t, (self.n, t.shape[-1]))
            tgt = np.full((self.n, 8), np.nan)
            tgt[:, :t.shape[1]] = t
        tol = np.ascontiguousarray(np.broadcast_to(np.asarray(tol, dtype=np.float64), (self.n,)).copy())
        mx = np.ascontiguousarray(np.broadcast_to(np.asarray(max_steps, dtype=np.int32), (self.n,)).copy())
        res, steps = np.zeros(self.n, dtype=np.int32), np.zeros(self.n, dtype=np.int32)
        self._check(self.lib.ur5_move_group(self._h, mask.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(tgt), _dp(tol), _ip(mx),
                                            _ip(res), _ip(steps)), "ur5_move_group")
        return res, steps

    def stay(self, ms):
        self._check(self.lib.ur5_stay(self._h, float(ms)), "ur5_stay")

    def move_ee(self, xyz, tol, max_steps):
        x = np.ascontiguousarray(np.broadcast_to(np.asarray(xyz, dtype=np.float64), (self.n, 3)).copy())
        tol = np.ascontiguousarray(np.broadcast_to(np.asarray(tol, dtype=np.float64), (self.n,)).copy())
        mx = np.ascontiguousarray(np.broadcast_to(np.asarray(max_steps, dtype=np.int32), (self.n,)).copy())
        res, steps = np.zeros(self.n, dtype=np.int32), np.zeros(self.n, dtype=np.int32)
        self._check(self.lib.ur5_move_ee(self._h, _dp(x), _dp(tol), _ip(mx), _ip(res), _ip(steps)), "ur5_move_ee")
        return res, steps

    def ik(self, xyz):
        x = np.ascontiguousarray(np.broadcast_to(np.asarray(xyz, dtype=np.float64), (self.n, 3)).copy())
        q5, res = np.zeros((self.n, 5)), np.zeros(self.n, dtype=np.int32)
        self._check(self.lib.ur5_ik(self._h, _dp(x), _dp(q5), _ip(res)), "ur5_ik")
        return q5, res

    def grasp_attempt(self, xyz, rot=0, check_mode=0, table_height=0.91, skip=None):
        a = np.zeros((self.n, 4))
        a[:, :3] = np.broadcast_to(np.asarray(xyz, dtype=np.float64), (self.n, 3))
        a[:, 3] = np.broadcast_to(np.asarray(rot, dtype=np.float64), (self.n,))
        rew = np.zeros(self.n, dtype=np.int32)
        ps, pr = np.zeros((self.n, 12), dtype=np.int32), np.zeros((self.n, 12), dtype=np.int32)
        sk = None if skip is None else np.ascontiguousarray(np.broadcast_to(np.asarray(skip), (self.n,)).astype(np.uint8))
        self._check(self.lib.ur5_grasp_attempt(self._h, _dp(a), None if sk is None else sk.ctypes.data_as(C.POINTER(C.c_uint8)), int(check_mode),
                                               float(table_height), _ip(rew), _ip(ps), _ip(pr)), "ur5_grasp_attempt")
        return rew, ps, pr

    def grasp_attempt_dev(self, action_ptr, reward_ptr, check_mode=0, table_height=0.91):
        """Asynchronous; action_ptr -> [n][8] float64, reward_ptr -> [n] int32, both HIP device pointers."""
        self._check(self.lib.ur5_grasp_attempt_dev(self._h, C.c_void_p(action_ptr), int(check_mode), float(table_height),
                                                   C.c_void_p(reward_ptr)), "ur5_grasp_attempt_dev")

    def grasp_attempt_reset_dev(self, action_ptr, reward_ptr, reset_seeds_ptr, check_mode=0, table_height=0.91, settle_ms=1000.0):
        """grasp_attempt_dev + reset_model (and its settle) for the scenes whose uint64 entry of reset_seeds_ptr is non-zero, one launch."""
        self._check(self.lib.ur5_grasp_attempt_reset_dev(self._h, C.c_void_p(action_ptr), int(check_mode), float(table_height), C.c_void_p(reward_ptr),
                                                         C.c_void_p(reset_seeds_ptr) if reset_seeds_ptr else None, float(settle_ms)),
                    "ur5_grasp_attempt_reset_dev")

    def grasp_rounds_dev(self, rule, round0, rounds, reward_ptr, action_out_ptr=None, check_mode=1, table_height=0.91, settle_ms=1000.0):
        """``rounds`` consecutive grasp rounds (+ episode resets) of every scene in ONE launch, aimed in the kernel by the scripted ``rule`` (an AimRule); no scene
        waits for another one's round. reward_ptr -> int32 [rounds][n], action_out_ptr -> float64 [rounds][n][8] or None (device pointers). Asynchronous."""
        self._check(self.lib.ur5_grasp_rounds_dev(self._h, C.byref(rule), int(round0), int(rounds), int(check_mode), float(table_height), C.c_void_p(reward_ptr),
                                                  C.c_void_p(action_out_ptr) if action_out_ptr else None, float(settle_ms)), "ur5_grasp_rounds_dev")

    def set_observation_dev(self, rgb_ptr, depth_ptr, camera, width, height, frames=1, depth_mode=0):
        """The following grasp_rounds_dev launches render every scene's observation at the start of each of its rounds into frame (round % frames) of
        rgb [frames][n][h][w][3] uint8 / depth [frames][n][h][w] float32 (device pointers, caller-owned); rgb_ptr None switches it off."""
        self._check(self.lib.ur5_set_observation_dev(self._h, int(camera), int(width), int(height), int(depth_mode), C.c_void_p(rgb_ptr) if rgb_ptr else None,
                                                     C.c_void_p(depth_ptr) if depth_ptr else None, int(frames)), "ur5_set_observation_dev")

    def model_uploads(self):
        """Test hook: model copies this handle's engine unit has sent to a device so far -- one per handle, at creation (include/ur5sim_test.h)."""
        return int(self.lib.ur5_model_uploads(self._h))

    def set_step_cap_dev(self, cap_ptr):
        """Test hook (include/ur5sim_test.h): int32 [n] device pointer of physics-step caps for the following grasp launches, or None."""
        self._check(self.lib.ur5_set_step_cap_dev(self._h, C.c_void_p(cap_ptr) if cap_ptr else None), "ur5_set_step_cap_dev")

    def sync(self):
        self._check(self.lib.ur5_sync(self._h), "ur5_sync")

    def set_stream(self, hip_stream):
        """Run on a caller-owned HIP stream given as an int handle, e.g. torch.cuda.current_stream().cuda_stream (0 = the default
        stream, which is torch's default); None = back to the handle's private stream."""
        if hip_stream is None:
            self._check(self.lib.ur5_set_stream(self._h, None, 0), "ur5_set_stream")
        else:
            self._check(self.lib.ur5_set_stream(self._h, C.c_void_p(int(hip_stream)) if hip_stream else None, 1), "ur5_set_stream")

    def set_order_dev(self, order_ptr):
        """Dispatch order of the following grasp / settle launches: device pointer to an int32 [n] permutation (caller keeps it alive); None = scene order."""
        self._check(self.lib.ur5_set_order_dev(self._h, C.c_void_p(order_ptr) if order_ptr else None), "ur5_set_order_dev")

    def set_order_view_dev(self, order_ptr):
        """The same without the handle's copy: the following launches read the caller's buffer, which must stay valid and unmodified until they have run."""
        self._check(self.lib.ur5_set_order_view_dev(self._h, C.c_void_p(order_ptr) if order_ptr else None), "ur5_set_order_view_dev")

    def last_launch_ms(self):
        return float(self.lib.ur5_last_launch_ms(self._h))

    def kernel_ms_total(self):
        return float(self.lib.ur5_kernel_ms_total(self._h))

    def state_device_ptr(self):
        return self.lib.ur5_state_device_ptr(self._h)

    def state_tensor(self, device):
        """The [n, stride] float64 state records (csrc/ur5_devmodel.h layout) as a torch tensor ALIASING the engine's device memory
        (host memory for the test-only emulation build): lets callers read object poses / write nothing without a PCIe round trip."""
        import torch
        stride = _VARIANT[self.variant][2]
        ptr = self.state_device_ptr()
        if torch.device(device).type == "cpu":
            buf = (C.c_double * (self.n * stride)).from_address(ptr)
            return torch.frombuffer(buf, dtype=torch.float64).view(self.n, stride)

        class _Alias:   # __cuda_array_interface__ v2: torch wraps the pointer without copying
            __cuda_array_interface__ = {"shape": (self.n, stride), "typestr": "<f8", "data": (int(ptr), False), "version": 2, "strides": None}
        return torch.as_tensor(_Alias(), device=device)

    def body_xpos(self):
        out = np.zeros((self.n, 8 + _VARIANT[self.variant][0], 3))
        self._check(self.lib.ur5_body_xpos(self._h, _dp(out)), "ur5_body_xpos")
        return out

    def render(self, camera_id=1, width=200, height=200, depth_mode=0):
        """rgb uint8 [n, h, w, 3], depth float32 [n, h, w] in get_image_data orientation (depth_mode 0 = metres, 1 = GL [0,1])."""
        rgb = np.zeros((self.n, height, width, 3), dtype=np.uint8)
        depth = np.zeros((self.n, height, width), dtype=np.float32)
        self._check(self.lib.ur5_render(self._h, int(camera_id), int(width), int(height), int(depth_mode),
                                        rgb.ctypes.data_as(C.POINTER(C.c_uint8)), depth.ctypes.data_as(C.POINTER(C.c_float))), "ur5_render")
        return rgb, depth

    def render_dev(self, rgb_ptr, depth_ptr, camera_id=1, width=200, height=200, depth_mode=0):
        self._check(self.lib.ur5_render_dev(self._h, int(camera_id), int(width), int(height), int(depth_mode), C.c_void_p(rgb_ptr),
                                            C.c_void_p(depth_ptr)), "ur5_render_dev")

    def forward_debug(self):
        maxobj, stride, _, maxcon = _VARIANT[self.variant]
        mb, mv = 8 + maxobj, 8 + 6 * maxobj
        out = np.zeros((self.n, stride))
        self._check(self.lib.ur5_forward_debug(self._h, _dp(out)), "ur5_forward_debug")
        o = 8
        d = dict(ncon=out[:, 0].astype(int), nsr=out[:, 1].astype(int))
        d["bpos"] = out[:, o:o + 3 * mb].reshape(self.n, mb, 3); o += 3 * mb
        d["Mr"] = out[:, o:o + 64].reshape(self.n, 8, 8); o += 64
        for k in ("qfrc_smooth", "qacc_smooth", "qacc"):
            d[k] = out[:, o:o + mv]; o += mv
        d["contacts"] = out[:, o:o + 10 * maxcon].reshape(self.n, maxcon, 10)
        if self.variant == 1:                                    # the pile engine: envelope size, and whether envelope / block cache of the step lived in the LDS pool
            d["envelope_doubles"] = out[:, 4].astype(int)
            d["env_in_lds"], d["dcache_in_lds"] = out[:, stride - 512 - 2].astype(int), out[:, stride - 512 - 1].astype(int)
        return d
