"""CompiledModel: the flat scene description shared by the CPU oracle and the HIP engine.

Plays the role of ``mujoco_py``'s ``PyMjModel`` [3P] for the fields the reference touches
(``MujocoController.py:36-43,249-254,737-758``): name<->id maps, ``actuator_trnid``, ``cam_*``,
``stat.extent``, ``vis.map``. Serialised as a "named-section" blob (``*.ur5m``):

    u64 magic 'UR5MODL1' | u64 nsections | nsections x { char name[32]; u32 dtype(0=f64,1=i32);
    u32 count; payload padded to 8 bytes } | trailing JSON (names) is kept in a side-car section.
"""
from __future__ import annotations

import json
import os
import struct

import numpy as np

GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 0, 2, 3, 5, 6, 7
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
MAGIC = b"UR5MODL1"

_ARRAYS = [
    "body_parentid", "body_pos", "body_quat", "body_jntadr", "body_jntnum", "body_dofadr", "body_dofnum",
    "body_weldid", "body_treeid", "body_mass", "body_ipos", "body_inertia", "body_invweight0",
    "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_pos", "jnt_axis", "jnt_limited", "jnt_range",
    "qpos0", "dof_bodyid", "dof_jntid", "dof_parentid", "dof_armature", "dof_damping", "dof_treeid",
    "dof_invweight0", "tree_dofadr", "tree_dofnum",
    "geom_type", "geom_bodyid", "geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_condim",
    "geom_margin", "geom_solref", "geom_solimp", "geom_rgba", "geom_meshid", "geom_rbound", "geom_collide",
    "mesh_vertadr", "mesh_vertnum", "mesh_vert", "vis_planeadr", "vis_planenum", "vis_plane", "pair_geom1", "pair_geom2",
    "eq_jnt1", "eq_jnt2", "eq_polycoef", "eq_solref", "eq_solimp",
    "act_jntid", "act_gear", "act_ctrlrange", "act_ctrllimited",
    "cam_pos", "cam_mat", "cam_fovy", "opt_f", "opt_i",
]


class CompiledModel:
    """Plain container; see ``mjcf.compile_mjcf`` for how the fields are produced."""

    def __init__(self):
        self.names = {}
        self.opt = {}

    # ---- sizes
    @property
    def nq(self):
        return len(self.qpos0)

    @property
    def nv(self):
        return len(self.dof_bodyid)

    @property
    def nu(self):
        return len(self.act_jntid)

    @property
    def nbody(self):
        return len(self.body_parentid)

    @property
    def ngeom(self):
        return len(self.geom_type)

    @property
    def ntree(self):
        return len(self.tree_dofadr)

    # ---- mujoco_py-style lookups used by the reference (MujocoController.py:249-254,341,488,745)
    def body_name2id(self, name):
        return self.names["body"].index(name)

    def joint_name2id(self, name):
        return self.names["joint"].index(name)

    def camera_name2id(self, name):
        return self.names["camera"].index(name)

    def actuator_id2name(self, i):
        return self.names["actuator"][i]

    def joint_id2name(self, i):
        return self.names["joint"][i]

    def get_joint_qpos_addr(self, name):
        j = self.joint_name2id(name)
        a = int(self.jnt_qposadr[j])
        n = {JNT_FREE: 7, JNT_BALL: 4}.get(int(self.jnt_type[j]), 1)
        return a if n == 1 else (a, a + n)

    @property
    def actuator_trnid(self):
        return np.stack([self.act_jntid, np.full_like(self.act_jntid, -1)], axis=1)

    @property
    def cam_pos0(self):
        return self.cam_pos

    @property
    def cam_mat0(self):
        return self.cam_mat

    # ---- serialisation
    def _pack_opt(self):
        o = self.opt
        self.opt_f = np.array([o["timestep"], o["tolerance"], o["impratio"], *o["gravity"], *o["jnt_solref"],
                               *o["jnt_solimp"], o["meaninertia"], o["extent"], o["znear"], o["zfar"]],
                              dtype=np.float64)
        self.opt_i = np.array([o["iterations"]], dtype=np.int32)

    def _unpack_opt(self):
        f = self.opt_f
        self.opt = dict(timestep=float(f[0]), tolerance=float(f[1]), impratio=float(f[2]),
                        gravity=[float(x) for x in f[3:6]], jnt_solref=[float(x) for x in f[6:8]],
                        jnt_solimp=[float(x) for x in f[8:13]], meaninertia=float(f[13]), extent=float(f[14]),
                        znear=float(f[15]), zfar=float(f[16]), iterations=int(self.opt_i[0]))

    def to_blob(self) -> bytes:
        self._pack_opt()
        parts = []
        for name in _ARRAYS:
            a = np.ascontiguousarray(getattr(self, name))
            if a.dtype.kind == "f":
                a, code = a.astype(np.float64), 0
            else:
                a, code = a.astype(np.int32), 1
            raw = a.tobytes()
            raw += b"\0" * ((-len(raw)) % 8)
            parts.append(struct.pack("<32sII", name.encode(), code, a.size) + raw)
        js = json.dumps(self.names).encode()
        js += b" " * ((-len(js)) % 8)
        parts.append(struct.pack("<32sII", b"names_json", 2, len(js)) + js)
        return MAGIC + struct.pack("<Q", len(parts)) + b"".join(parts)

    @classmethod
    def from_blob(cls, blob: bytes) -> "CompiledModel":
        if blob[:8] != MAGIC:
            raise ValueError("not a UR5MODL1 blob")
        m = cls()
        n = struct.unpack_from("<Q", blob, 8)[0]
        off = 16
        for _ in range(n):
            name, code, count = struct.unpack_from("<32sII", blob, off)
            off += 40
            name = name.rstrip(b"\0").decode()
            if code == 2:
                m.names = json.loads(blob[off:off + count].decode())
                off += count
                continue
            dt = np.float64 if code == 0 else np.int32
            nbytes = count * np.dtype(dt).itemsize
            setattr(m, name, np.frombuffer(blob, dtype=dt, count=count, offset=off).copy())
            off += nbytes + ((-nbytes) % 8)
        m._reshape()
        m._unpack_opt()
        return m

    _SHAPES = dict(body_pos=3, body_quat=4, body_ipos=3, body_inertia=6, body_invweight0=2, jnt_pos=3, jnt_axis=3,
                   jnt_range=2, geom_size=3, geom_pos=3, geom_quat=4, geom_friction=3, geom_solref=2, geom_solimp=5,
                   geom_rgba=4, mesh_vert=3, vis_plane=4, eq_polycoef=5, eq_solref=2, eq_solimp=5, act_ctrlrange=2, cam_pos=3,
                   cam_mat=9)

    def _reshape(self):
        for k, w in self._SHAPES.items():
            setattr(self, k, getattr(self, k).reshape(-1, w))

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.to_blob())

    @classmethod
    def load(cls, path):
        with open(path, "rb") as f:
            return cls.from_blob(f.read())


ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

# file= kwarg of the reference's GraspEnv (GraspingEnv.py:30) -> shipped compiled asset
KNOWN_MODELS = {
    "/UR5+gripper/UR5gripper_2_finger.xml": "ur5_2f.ur5m",
    "/UR5+gripper/UR5gripper_2_finger_many_objects.xml": "ur5_2f_many.ur5m",
    "it1_4box": "ur5_2f_it1_4box.ur5m",
    "many_objects_arm_collision": "ur5_2f_many_armcol.ur5m",   # the 40-object scene with the arm-link hulls colliding
}


def load_model(spec: str) -> CompiledModel:
    """Load a model by reference file name (shipped pre-compiled), asset alias, ``.ur5m`` or ``.xml`` path."""
    if spec in KNOWN_MODELS:
        return CompiledModel.load(os.path.join(ASSET_DIR, KNOWN_MODELS[spec]))
    if spec.endswith(".ur5m"):
        return CompiledModel.load(spec)
    if spec.endswith(".xml"):
        from .mjcf import compile_mjcf
        return compile_mjcf(spec)
    raise ValueError(f"unknown model spec {spec!r}")
