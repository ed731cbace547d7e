"""Batched ``MJ_Controller``: the reference's robot-controller surface on top of the HIP engine.

Mirrors ``gym_grasper/controller/MujocoController.py`` (class ``MJ_Controller``, :21): same method names, keyword
arguments and result strings, but every call acts on ``n_envs`` independent scenes at once. With ``n_envs == 1`` the
return values have the reference's scalar shapes (one string, one array); otherwise lists / arrays gain a leading N.

Not carried over (out of scope): viewer markers, joint-angle plots, debug printing, ``ik_2``,
``toss_it_from_the_ellbow``, the ``show`` window of ``get_image_data``.
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np

from .model import CompiledModel, load_model
from .native import BatchSim, RES_SUCCESS, RES_MAX_STEPS, RES_IK_FAIL

IK_FAIL_STRING = "No valid joint angles received, could not move EE to position."  # MujocoController.py:463


def result_string(code, max_steps):
    """Engine result code -> the reference's strings (MujocoController.py:362,376,463)."""
    if code == RES_SUCCESS:
        return "success"
    if code == RES_MAX_STEPS:
        return "max. steps reached: {}".format(max_steps)
    if code == RES_IK_FAIL:
        return IK_FAIL_STRING
    return ""


class MJ_Controller(object):
    def __init__(self, model=None, simulation=None, viewer=None, n_envs=1, device_id=0, _lib_path=None):
        if model is None:
            model = "/UR5+gripper/UR5gripper_2_finger.xml"           # MujocoController.py:33
        self.model = model if isinstance(model, CompiledModel) else load_model(model)
        self.sim = simulation if simulation is not None else BatchSim(self.model, n_envs, device_id=device_id, lib_path=_lib_path)
        self.n_envs = self.sim.n
        self.viewer = viewer                                          # kept for signature parity; never used
        nu = self.model.nu
        self.groups = defaultdict(list)
        self.groups["All"] = list(range(nu))                          # :39
        self.create_group("Arm", list(range(5)))                      # :41
        self.create_group("Gripper", [6])                             # :42
        self.actuated_joint_ids = np.array(self.model.act_jntid)      # :43
        self.actuators = [[i, self.model.actuator_id2name(i), int(self.model.act_jntid[i]),
                           self.model.joint_id2name(int(self.model.act_jntid[i]))] for i in range(nu)]
        self.reached_target = False
        self.cam_matrix = None
        self.cam_init = False
        self.last_movement_steps = 0
        self.current_carthesian_target = None

    # ------------------------------------------------------------------ helpers
    def _one(self, v):
        return v[0] if self.n_envs == 1 else v

    @property
    def current_target_joint_values(self):
        """Per-env setpoints of the 7 PIDs (MujocoController.py:236-241)."""
        return self._one(self.sim.get_state()["pid"][:, :, 0])

    @property
    def qpos(self):
        return self._one(self.sim.get_state()["qpos"])

    @property
    def last_steps(self):                                             # :827-829
        return self.last_movement_steps

    def _mask(self, group):
        m = 0
        for i in self.groups[group]:
            m |= 1 << i
        return m

    # ------------------------------------------------------------------ groups / low level
    def create_group(self, group_name, idx_list):                     # :53-77
        try:
            assert len(idx_list) <= self.model.nu, "Too many joints specified!"
            assert group_name not in self.groups.keys(), "A group with name {} already exists!".format(group_name)
            assert np.max(idx_list) <= self.model.nu, "List contains invalid actuator ID (too high)"
            self.groups[group_name] = idx_list
        except Exception as e:
            print(e)
            print("Could not create a new group.")

    def actuate_joint_group(self, group, motor_values):               # :256-267
        try:
            assert group in self.groups.keys(), "No group with name {} exists!".format(group)
            mv = np.atleast_2d(np.asarray(motor_values, dtype=np.float64))
            assert mv.shape[-1] == len(self.groups[group]), "Invalid number of actuator values!"
            ctrl = self.sim.get_ctrl()
            ctrl[:, self.groups[group]] = np.broadcast_to(mv, (self.n_envs, mv.shape[-1]))
            self.sim.set_ctrl(ctrl)
        except Exception as e:
            print(e)
            print("Could not actuate requested joint group.")

    def step(self, n=1):
        """``sim.step()`` x n with the motor values last written (MujocoController.py:379)."""
        self.sim.step(n)

    def set_group_joint_target(self, group, target):                  # :395-406
        idx = self.groups[group]
        try:
            t = np.atleast_2d(np.asarray(target, dtype=np.float64))
            assert t.shape[-1] == len(idx), "Length of the target must match the number of actuated joints in the group."
            st = self.sim.get_state()["pid"]
            st[:, idx, 0] = np.broadcast_to(t, (self.n_envs, len(idx)))
            self.sim.set_state(pid=st)
        except Exception as e:
            print(e)
            print(f"Could not set new group joint target for group {group}")

    def move_group_to_joint_target(self, group="All", target=None, tolerance=0.1, max_steps=10000, plot=False, marker=False,
                                   render=True, quiet=False):         # :269-393
        try:
            assert group in self.groups.keys(), "No group with name {} exists!".format(group)
            if target is not None:
                t = np.atleast_2d(np.asarray(target, dtype=np.float64))
                assert t.shape[-1] == len(self.groups[group]), "Mismatching target dimensions for group {}!".format(group)
                target = t
            res, steps = self.sim.move_group(self._mask(group), target, tolerance, max_steps)
            self.last_movement_steps = self._one(steps)
            self.reached_target = bool(np.all(res == RES_SUCCESS))
            return self._one([result_string(int(r), max_steps) for r in res])
        except Exception as e:
            print(e)
            print("Could not move to requested joint target.")

    def open_gripper(self, half=False, **kwargs):                     # :408-421
        kwargs = {k: v for k, v in kwargs.items() if k in ("plot", "marker", "render", "quiet")}
        return self.move_group_to_joint_target(group="Gripper", target=[0.0 if half else 0.4], max_steps=1000, tolerance=0.05, **kwargs)

    def close_gripper(self, **kwargs):                                # :423-434
        return self.move_group_to_joint_target(group="Gripper", target=[-0.4], tolerance=0.01, **kwargs)

    def grasp(self, **kwargs):                                        # :436-444
        result = self.close_gripper(max_steps=300, **kwargs)
        if self.n_envs == 1:
            return result != "success"
        return [r != "success" for r in result]

    def move_ee(self, ee_position, **kwargs):                         # :446-465
        xyz = np.atleast_2d(np.asarray(ee_position, dtype=np.float64))
        self.current_carthesian_target = xyz.copy()
        tol = kwargs.get("tolerance", 0.1)
        max_steps = kwargs.get("max_steps", 10000)
        res, steps = self.sim.move_ee(xyz, tol, max_steps)
        self.last_movement_steps = self._one(steps)
        return self._one([result_string(int(r), max_steps) for r in res])

    def ik(self, ee_position):                                        # :467-517
        try:
            xyz = np.atleast_2d(np.asarray(ee_position, dtype=np.float64))
            assert xyz.shape[-1] == 3, "Invalid EE target! Please specify XYZ-coordinates in a list of length 3."
            q5, res = self.sim.ik(xyz)
            out = [q5[e] if res[e] == RES_SUCCESS else None for e in range(self.n_envs)]
            if any(o is None for o in out):
                print("Failed to find IK solution.")
            return self._one(out)
        except Exception as e:
            print(e)
            print("Could not find an inverse kinematics solution.")

    def stay(self, duration, render=True):                            # :621-636 (deterministic 10-step chunks)
        self.sim.stay(duration)

    # ------------------------------------------------------------------ camera maths (host side, per grasp -- not per step)
    def get_image_data(self, show=False, camera="top_down", width=200, height=200):   # :708-727
        """RGB (uint8) and GL window depth in [0, 1] (float32) of ``camera``, both flips applied -- what the reference returns;
        feed the depth to :meth:`depth_2_meters`. Rendered on the GPU by ray casting (csrc/ur5_raster.h)."""
        rgb, depth = self.sim.render(self.model.camera_name2id(camera), width, height, depth_mode=1)
        return self._one(rgb), self._one(depth)

    def depth_2_meters(self, depth):                                  # :729-740
        extend = self.model.opt["extent"]
        near = self.model.opt["znear"] * extend
        far = self.model.opt["zfar"] * extend
        return near / (1 - depth * (1 - near / far))

    def create_camera_data(self, width, height, camera):              # :742-759
        cam_id = self.model.camera_name2id(camera)
        fovy = self.model.cam_fovy[cam_id]
        f = 0.5 * height / np.tan(fovy * np.pi / 360)
        self.cam_matrix = np.array(((f, 0, width / 2), (0, f, height / 2), (0, 0, 1)))
        self.cam_rot_mat = np.reshape(self.model.cam_mat0[cam_id], (3, 3))
        self.cam_pos = self.model.cam_pos0[cam_id]
        self.cam_init = True

    def world_2_pixel(self, world_coordinate, width=200, height=200, camera="top_down"):   # :761-781
        if not self.cam_init:
            self.create_camera_data(width, height, camera)
        hom_pixel = self.cam_matrix @ self.cam_rot_mat @ (np.asarray(world_coordinate) - self.cam_pos)
        pixel = hom_pixel[:2] / hom_pixel[2]
        return np.round(pixel[0]).astype(int), np.round(pixel[1]).astype(int)

    def pixel_2_world_batch(self, pixel_x, pixel_y, depth, width=200, height=200, camera="top_down"):
        """pixel_2_world (:783-806) for arrays of pixels at once -> [N, 3]; same arithmetic, no Python loop over the scenes."""
        if not self.cam_init:
            self.create_camera_data(width, height, camera)
        px = np.stack([np.asarray(pixel_x, dtype=np.float64), np.asarray(pixel_y, dtype=np.float64), np.ones(len(np.atleast_1d(pixel_x)))])
        pos_c = np.linalg.inv(self.cam_matrix) @ (px * (-np.asarray(depth, dtype=np.float64)))
        return (np.linalg.inv(self.cam_rot_mat) @ (pos_c + np.asarray(self.cam_pos)[:, None])).T

    def pixel_2_world(self, pixel_x, pixel_y, depth, width=200, height=200, camera="top_down"):   # :783-806
        if not self.cam_init:
            self.create_camera_data(width, height, camera)
        pixel_coord = np.array([pixel_x, pixel_y, 1]) * (-depth)
        pos_c = np.linalg.inv(self.cam_matrix) @ pixel_coord
        pos_w = np.linalg.inv(self.cam_rot_mat) @ (pos_c + self.cam_pos)
        return pos_w
