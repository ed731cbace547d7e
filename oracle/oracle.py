"""ctypes wrapper around libur5_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg. Nothing under
mujoco_rl_ur5_amd/ imports this module (tests/test_layout.py enforces it).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

RES_SUCCESS, RES_MAX_STEPS, RES_IK_FAIL = 0, 1, 2


def build(force=False, variant=""):
    """variant "" = the oracle (no contraction); "fma" = the same text compiled with fused multiply-adds (-ffp-contract=fast -mfma): the independent-arithmetic
    control twin of tools/pile_divergence_time.py (round-5 verdict 1b). Never the checker of a parity test."""
    name = "libur5_oracle%s.so" % ("_" + variant if variant else "")
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "ur5_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", name])
    return so


_LIBS = {}


def lib(variant=""):
    if variant not in _LIBS:
        _LIBS[variant] = _load(build(variant=variant))
    return _LIBS[variant]


def _load(so):
    L = C.CDLL(so)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    L.ur5o_create.restype = vp
    L.ur5o_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int]
    L.ur5o_destroy.argtypes = [vp]
    for f in ("ur5o_nq", "ur5o_nv", "ur5o_nu", "ur5o_ncon", "ur5o_nefc", "ur5o_solver_iter_last", "ur5o_last_steps"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_int
    for f in ("ur5o_total_steps", "ur5o_solver_iters", "ur5o_bad_state_resets"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_long
    L.ur5o_set_options.argtypes = [vp, C.c_int, C.c_double, C.c_int]
    L.ur5o_set_solver_limits.argtypes = [vp, C.c_int, C.c_double]
    L.ur5o_set_contact_order.argtypes = [vp, C.c_int]
    L.ur5o_primal_cost.argtypes = [vp, dp, dp, dp]
    L.ur5o_get_state.argtypes = [vp, dp, dp, dp, dp]
    L.ur5o_set_state.argtypes = [vp, dp, dp, dp, dp]
    L.ur5o_set_ctrl.argtypes = [vp, dp]
    L.ur5o_get_ctrl.argtypes = [vp, dp]
    L.ur5o_forward.argtypes = [vp]
    L.ur5o_set_cholesky_order.argtypes = [vp, C.c_int]
    L.ur5o_set_checkpoints.argtypes = [vp, ip, C.c_int]
    L.ur5o_get_checkpoints.argtypes = [vp, dp]
    L.ur5o_get_checkpoints.restype = C.c_int
    L.ur5o_bench_pile_aim.argtypes = [vp, C.c_int, C.c_int, dp]
    L.ur5o_bench_pile_aim.restype = C.c_int
    L.ur5o_newton_trace.argtypes = [vp, C.c_int]
    L.ur5o_get_newton_trace.argtypes = [vp, vp, C.c_int]
    L.ur5o_get_newton_trace.restype = C.c_int
    L.ur5o_get_row_contacts.argtypes = [vp, vp]
    L.ur5o_step.argtypes = [vp, C.c_int]
    L.ur5o_reset.argtypes = [vp, C.c_uint64, C.c_int, C.c_int]
    L.ur5o_move_group.argtypes = [vp, C.c_uint, dp, C.c_double, C.c_int, ip]
    L.ur5o_move_group.restype = C.c_int
    L.ur5o_move_group_plot.argtypes = [vp, C.c_uint, dp, C.c_double, C.c_int, C.c_int, C.c_int, ip, dp, ip, ip]
    L.ur5o_move_group_plot.restype = C.c_int
    L.ur5o_stay.argtypes = [vp, C.c_double]
    L.ur5o_ik.argtypes = [vp, dp, dp]
    L.ur5o_ik.restype = C.c_int
    L.ur5o_move_ee.argtypes = [vp, dp, C.c_double, C.c_int, ip]
    L.ur5o_move_ee.restype = C.c_int
    L.ur5o_open_gripper.argtypes = [vp, C.c_int]
    L.ur5o_open_gripper.restype = C.c_int
    L.ur5o_close_gripper.argtypes = [vp, C.c_int]
    L.ur5o_close_gripper.restype = C.c_int
    L.ur5o_grasp_attempt.argtypes = [vp, dp, C.c_int, C.c_int, C.c_double, ip, ip]
    L.ur5o_grasp_attempt.restype = C.c_int
    L.ur5o_body_xpos.argtypes = [vp, dp]
    L.ur5o_body_xmat.argtypes = [vp, dp]
    L.ur5o_mass_matrix.argtypes = [vp, dp]
    L.ur5o_get_vec.argtypes = [vp, C.c_int, dp]
    L.ur5o_get_contacts.argtypes = [vp, dp]
    L.ur5o_get_rows.argtypes = [vp, dp]
    L.ur5o_render.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.POINTER(C.c_float)]
    L.ur5o_batch_camera.argtypes = [C.c_int]
    L.ur5o_batch.restype = C.c_long
    L.ur5o_batch.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_double),
                             C.POINTER(C.c_long), C.POINTER(C.c_long)]
    return L


def batch(model, nthreads, budget_s, mode=0, nsteps=100):
    """bench.py's CPU baseline: `nthreads` native threads, one scene each at a time, for `budget_s` seconds of wall time
    (mode 0 = IT1 reset + one aimed grasp attempt per scene, mode 1 = the first `nsteps` steps of the many-object drop,
    mode 2 = bench.py's stationary IT1 workload: episodes of reset + settle + `nsteps` aimed attempts, mode 3 = the same episodes with a rendered
    observation and depth-derived grasp height per attempt (bench.py kind "it4"), mode 4 = 40-object piles: reset + settle + one rendered attempt).
    Returns (physics steps, scenes completed, wall seconds, grasp attempts, successes)."""
    lib().ur5o_batch_camera(int(model.camera_name2id("top_down")))
    blob = model.to_blob()
    scenes, wall, att, suc = C.c_long(0), C.c_double(0), C.c_long(0), C.c_long(0)
    steps = lib().ur5o_batch(blob, len(blob), model.body_name2id("ee_link"), model.body_name2id("base_link"), int(nthreads), float(budget_s),
                             int(mode), int(nsteps), C.byref(scenes), C.byref(wall), C.byref(att), C.byref(suc))
    return int(steps), int(scenes.value), float(wall.value), int(att.value), int(suc.value)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


class Oracle:
    """One fp64 scene. Mirrors the C functions 1:1; arrays are numpy float64."""

    def __init__(self, model, variant=""):
        self.model = model
        self._L = lib(variant)
        blob = model.to_blob()
        self._h = self._L.ur5o_create(blob, len(blob), model.body_name2id("ee_link"), model.body_name2id("base_link"))
        if not self._h:
            raise RuntimeError("oracle rejected the model blob")
        self.nq, self.nv, self.nu = model.nq, model.nv, model.nu

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.ur5o_destroy(self._h)
            self._h = None

    def set_options(self, contacts_enabled=1, pid_dt=0.0, solver=0):
        """solver: 0 = Newton (default), 1 = PGS."""
        self._L.ur5o_set_options(self._h, contacts_enabled, pid_dt, solver)

    def set_solver_limits(self, iterations=0, tolerance=-1.0):
        """Override the model's solver iteration cap / tolerance (0 / negative: the model's own)."""
        self._L.ur5o_set_solver_limits(self._h, int(iterations), float(tolerance))

    def set_contact_order(self, mode):
        """Test hook: 0 = contacts in geom-pair order, 1 = the same contacts reversed (a rounding-level perturbation of every sum over contacts)."""
        self._L.ur5o_set_contact_order(self._h, int(mode))

    def primal_cost(self, qacc):
        """(cost, |gradient|) of the constraint QP of the last forward() at the acceleration `qacc`."""
        x = np.ascontiguousarray(qacc, dtype=np.float64)
        c, g = C.c_double(0), C.c_double(0)
        self._L.ur5o_primal_cost(self._h, _dp(x), C.byref(c), C.byref(g))
        return c.value, g.value

    def get_state(self):
        qpos, qvel, warm, pid = np.zeros(self.nq), np.zeros(self.nv), np.zeros(self.nv), np.zeros((self.nu, 4))
        self._L.ur5o_get_state(self._h, _dp(qpos), _dp(qvel), _dp(warm), _dp(pid))
        return dict(qpos=qpos, qvel=qvel, warmstart=warm, pid=pid)

    def set_state(self, qpos=None, qvel=None, warmstart=None, pid=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (qpos, qvel, warmstart, pid)]
        self._L.ur5o_set_state(self._h, *[_dp(a) for a in arrs])

    @property
    def qpos(self):
        return self.get_state()["qpos"]

    @property
    def qvel(self):
        return self.get_state()["qvel"]

    def set_ctrl(self, ctrl):
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        self._L.ur5o_set_ctrl(self._h, _dp(c))

    def get_ctrl(self):
        c = np.zeros(self.nu)
        self._L.ur5o_get_ctrl(self._h, _dp(c))
        return c

    def forward(self):
        self._L.ur5o_forward(self._h)

    def step(self, n=1):
        self._L.ur5o_step(self._h, n)

    def reset(self, seed, mode=1, settle=True):
        self._L.ur5o_reset(self._h, seed, mode, int(settle))

    def move_group(self, mask, target, tol, max_steps):
        steps = C.c_int(0)
        t = None if target is None else np.ascontiguousarray(target, dtype=np.float64)
        r = self._L.ur5o_move_group(self._h, mask, _dp(t), tol, max_steps, C.byref(steps))
        return r, steps.value

    def move_group_plot(self, mask, target, tol, max_steps, every=2, cap=4096):
        """move_group with the reference's plot=True recording: (result, steps, plot_steps[n], plot_q[n, group size])."""
        nj = bin(mask).count("1")
        ps, pq = np.zeros(cap, dtype=np.int32), np.zeros((cap, nj))
        n, steps = C.c_int(0), C.c_int(0)
        t = np.ascontiguousarray(target, dtype=np.float64)
        r = self._L.ur5o_move_group_plot(self._h, mask, _dp(t), tol, max_steps, every, cap, ps.ctypes.data_as(C.POINTER(C.c_int)), _dp(pq),
                                       C.byref(n), C.byref(steps))
        return r, steps.value, ps[:n.value], pq[:n.value]

    def stay(self, ms):
        self._L.ur5o_stay(self._h, float(ms))

    def ik(self, xyz):
        out = np.zeros(5)
        x = np.ascontiguousarray(xyz, dtype=np.float64)
        ok = self._L.ur5o_ik(self._h, _dp(x), _dp(out))
        return bool(ok), out

    def move_ee(self, xyz, tol, max_steps):
        steps = C.c_int(0)
        x = np.ascontiguousarray(xyz, dtype=np.float64)
        r = self._L.ur5o_move_ee(self._h, _dp(x), tol, max_steps, C.byref(steps))
        return r, steps.value

    def open_gripper(self, half=False):
        return self._L.ur5o_open_gripper(self._h, int(half))

    def close_gripper(self, max_steps):
        return self._L.ur5o_close_gripper(self._h, max_steps)

    def grasp_attempt(self, xyz, rot=0, check_mode=0, table_height=0.91):
        ps, pr = np.zeros(12, dtype=np.int32), np.zeros(12, dtype=np.int32)
        x = np.ascontiguousarray(xyz, dtype=np.float64)
        r = self._L.ur5o_grasp_attempt(self._h, _dp(x), rot, check_mode, table_height,
                                     ps.ctypes.data_as(C.POINTER(C.c_int)), pr.ctypes.data_as(C.POINTER(C.c_int)))
        return r, ps, pr

    @property
    def total_steps(self):
        return self._L.ur5o_total_steps(self._h)

    @property
    def bad_state_resets(self):
        """How often a step produced a non-finite / > 1e10 state and the scene went back to qpos0 (mj_resetData [3P])."""
        return self._L.ur5o_bad_state_resets(self._h)

    @property
    def solver_iters(self):
        return self._L.ur5o_solver_iters(self._h)

    @property
    def last_steps(self):
        return self._L.ur5o_last_steps(self._h)

    def body_xpos(self):
        out = np.zeros((self.model.nbody, 3))
        self._L.ur5o_body_xpos(self._h, _dp(out))
        return out

    def body_xmat(self):
        out = np.zeros((self.model.nbody, 3, 3))
        self._L.ur5o_body_xmat(self._h, _dp(out))
        return out

    def mass_matrix(self):
        out = np.zeros((self.nv, self.nv))
        self._L.ur5o_mass_matrix(self._h, _dp(out))
        return out

    def vec(self, name):
        which = ["qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc_smooth", "qacc", "qfrc_constraint"].index(name)
        out = np.zeros(self.nv)
        self._L.ur5o_get_vec(self._h, which, _dp(out))
        return out

    def render(self, camera_id=1, width=200, height=200, depth_mode=0):
        rgb = np.zeros((height, width, 3), dtype=np.uint8)
        depth = np.zeros((height, width), dtype=np.float32)
        self._L.ur5o_render(self._h, camera_id, width, height, depth_mode, rgb.ctypes.data_as(C.POINTER(C.c_ubyte)),
                          depth.ctypes.data_as(C.POINTER(C.c_float)))
        return rgb, depth

    def contacts(self):
        n = self._L.ur5o_ncon(self._h)
        out = np.zeros((max(n, 1), 12))
        self._L.ur5o_get_contacts(self._h, _dp(out))
        return out[:n]

    def rows(self):
        n = self._L.ur5o_nefc(self._h)
        out = np.zeros((max(n, 1), 6))
        self._L.ur5o_get_rows(self._h, _dp(out))
        return out[:n]

    @property
    def solver_iter_last(self):
        return self._L.ur5o_solver_iter_last(self._h)

    def set_cholesky_order(self, mode):
        """Test hook: 1 = the Newton solve eliminates the dofs in reversed order (same mathematics, another rounding: a 'different text' twin)."""
        self._L.ur5o_set_cholesky_order(self._h, int(mode))

    def set_checkpoints(self, steps):
        """Record qpos after these numbers of steps (ascending), counted from now (test hook of ur5_oracle.cpp Sim::step)."""
        a = np.ascontiguousarray(steps, dtype=np.int32)
        self._L.ur5o_set_checkpoints(self._h, a.ctypes.data_as(C.POINTER(C.c_int)), len(a))

    def get_checkpoints(self):
        """qpos [k, nq] of the checkpoints reached so far."""
        k = self._L.ur5o_get_checkpoints(self._h, None)
        out = np.zeros((max(k, 1), self.model.nq))
        self._L.ur5o_get_checkpoints(self._h, _dp(out))
        return out[:k]

    def bench_pile_aim(self, g, r=0):
        """bench.py's pile aiming rule on the oracle's current state (ur5_oracle.cpp bench_pile_aim): (xy, rotation index)."""
        xyz = np.zeros(3)
        rot = self._L.ur5o_bench_pile_aim(self._h, int(g), int(r), _dp(xyz))
        return xyz[:2].copy(), int(rot)

    def newton_trace(self, on=True):
        """Switch the recording of the Newton solve's active sets on / off (test hook of ur5_oracle.cpp newton_direction)."""
        self._L.ur5o_newton_trace(self._h, 1 if on else 0)

    def get_newton_trace(self):
        """(active uint8 [evaluations, rows], contact index of every row int32 [rows], -1 for equality / limit rows) of the LAST solve."""
        ne = self._L.ur5o_nefc(self._h)
        n = self._L.ur5o_get_newton_trace(self._h, None, 0)
        out = np.zeros((max(n, 1), max(ne, 1)), dtype=np.uint8)
        self._L.ur5o_get_newton_trace(self._h, out.ctypes.data_as(C.c_void_p), n)
        rc = np.zeros(max(ne, 1), dtype=np.int32)
        self._L.ur5o_get_row_contacts(self._h, rc.ctypes.data_as(C.c_void_p))
        return out[:n, :ne], rc[:ne]
