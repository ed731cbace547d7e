"""MJCF-subset model compiler for the UR5 + 2-finger-gripper pick scenes.

Turns the reference's scene descriptions (``UR5+gripper/UR5gripper_2_finger.xml``,
``UR5gripper_2_finger_many_objects.xml`` + ``objects.xml``) into a flat
:class:`~mujoco_rl_ur5_amd.model.CompiledModel` that both the CPU oracle and the HIP engine load.

It replaces what the reference gets from ``mujoco_py.load_model_from_path`` [3P]
(``gym_grasper/controller/MujocoController.py:33``, ``gym_grasper/envs/GraspingEnv.py:47``).
Only the MJCF features those two files use are understood; anything else raises.

Documented deviations from a real MuJoCo compile (see DESIGN.md "Model constants"):
  * mesh geoms keep the mesh's own frame (no re-centring to the principal axes); collision uses the
    convex hull of the vertices, as MuJoCo does;
  * mesh inertia: default ``mesh_inertia="dedup"`` = exact signed-tetrahedra volume of the
    triangle set with repeated triangles removed (every UR5 visual STL stores each triangle twice), density
    1000. With it the reference's scripted moves converge the way media/console.png records (all phases
    "success"); the double-counted ``"signed"`` / ``"legacy"`` (|volume| pyramids, MuJoCo <= 2.1 as recalled)
    variants leave a P-only steady-state error above the 0.01 rad tolerance. Kept selectable;
  * the seven UR5 arm-link meshes are not collidable unless ``arm_collision=True``;
  * collision hulls of the gripper meshes are the FULL hulls of the STLs (400 / 70 / 120 vertices for
    robotiq_85_base_link_coarse / inner_knuckle_coarse / inner_finger_coarse), as in the reference
    (``UR5gripper_2_finger.xml:54-71,188-212``): ``maxhullvert=0``. Rounds 1-2 capped them at 32 vertices; on the oracle that
    changed 37 of 240 reward bits of the IT1 scene (``tools/hull_cap_effect.py``, ``profiles/r03_hull_cap_effect.json``).
    A cap > 0 grows the hull farthest-point-first (MuJoCo's ``maxhullvert`` [3P, 3.x]) and stays available for experiments;
    the ray caster's face lists stay capped (``maxvisvert``) -- they are a rendering proxy, not collision geometry.
"""
from __future__ import annotations

import math
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np

from .model import CompiledModel, GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH
from .model import JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE

_GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
               "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH}
_JNT_TYPES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}
_ARM_MESHES = {"base", "shoulder", "upperarm", "forearm", "wrist1", "wrist2", "wrist3"}


# ----------------------------------------------------------------------------- small maths
def _floats(s, n=None):
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and len(v) != n:
        raise ValueError(f"expected {n} numbers, got {s!r}")
    return v


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def axisangle_to_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < 1e-14 or angle == 0.0:
        return np.array([1.0, 0, 0, 0])
    axis = axis / n
    s = math.sin(angle / 2)
    return np.array([math.cos(angle / 2), *(axis * s)])


def euler_to_quat(e):
    # MuJoCo default eulerseq "xyz" (intrinsic rotations about x, then y, then z)
    q = np.array([1.0, 0, 0, 0])
    for i, a in enumerate(e):
        ax = np.zeros(3)
        ax[i] = 1.0
        q = quat_mul(q, axisangle_to_quat(ax, a))
    return q


def _orientation(el):
    if "quat" in el.attrib:
        q = _floats(el.get("quat"), 4)
        return q / np.linalg.norm(q)
    if "axisangle" in el.attrib:
        v = _floats(el.get("axisangle"), 4)
        return axisangle_to_quat(v[:3], v[3])
    if "euler" in el.attrib:
        return euler_to_quat(_floats(el.get("euler"), 3))
    return np.array([1.0, 0, 0, 0])


# ----------------------------------------------------------------------------- meshes
def load_stl(path):
    """Binary STL -> (ntri, 3, 3) float64 triangle array."""
    with open(path, "rb") as f:
        data = f.read()
    ntri = struct.unpack_from("<I", data, 80)[0]
    if len(data) < 84 + 50 * ntri:
        raise ValueError(f"{path}: not a binary STL")
    rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                        count=ntri, offset=84)
    return rec["v"].astype(np.float64)


def mesh_inertia_legacy(tris, density):
    """|volume| pyramids about the area-weighted surface centroid ([3P] recollection)."""
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    cen = ((a + b + c) / 3.0 * area[:, None]).sum(0) / area.sum()
    a, b, c = a - cen, b - cen, c - cen
    vol = np.abs(np.einsum("ij,ij->i", a, np.cross(b, c))) / 6.0
    mass = density * vol.sum()
    # tetra (0,a,b,c): com = (a+b+c)/4 ; second moments via canonical formula
    com = ((a + b + c) / 4.0 * vol[:, None]).sum(0) / vol.sum()
    # covariance integral of a tetra with apex at origin: vol/20 * (sum_i v_i v_i^T + (sum v)(sum v)^T)
    s = a + b + c
    cov = (np.einsum("n,ni,nj->ij", vol, a, a) + np.einsum("n,ni,nj->ij", vol, b, b)
           + np.einsum("n,ni,nj->ij", vol, c, c) + np.einsum("n,ni,nj->ij", vol, s, s)) / 20.0
    cov *= density
    inertia_o = np.trace(cov) * np.eye(3) - cov          # about the centroid `cen`
    d = com
    inertia_c = inertia_o - mass * (d.dot(d) * np.eye(3) - np.outer(d, d))
    return mass, com + cen, inertia_c


def mesh_inertia_signed(tris, density, dedup=False):
    """Signed-tetrahedra (exact for a closed, consistently wound shell). ``dedup`` first drops repeated triangles:
    every UR5 visual STL of the reference stores each triangle twice (SURVEY.md H1 measured 2x volumes)."""
    if dedup:
        key = np.sort(np.round(tris.reshape(-1, 3, 3), 7).reshape(len(tris), 3, 3).view([("", float)] * 3).reshape(len(tris), 3), axis=1)
        _, keep = np.unique(key, axis=0, return_index=True)
        tris = tris[np.sort(keep)]
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)) / 6.0
    tot = vol.sum()
    mass = density * tot
    com = ((a + b + c) / 4.0 * vol[:, None]).sum(0) / tot
    s = a + b + c
    cov = (np.einsum("n,ni,nj->ij", vol, a, a) + np.einsum("n,ni,nj->ij", vol, b, b)
           + np.einsum("n,ni,nj->ij", vol, c, c) + np.einsum("n,ni,nj->ij", vol, s, s)) / 20.0 * density
    inertia_o = np.trace(cov) * np.eye(3) - cov
    inertia_c = inertia_o - mass * (com.dot(com) * np.eye(3) - np.outer(com, com))
    return mass, com, inertia_c


def convex_hull_vertices(verts, maxhullvert=0):
    """Hull vertices of a mesh; with ``maxhullvert`` > 0 the hull is grown greedily (farthest point first, the order
    quickhull adds points) and stopped at that many vertices -- MuJoCo's ``maxhullvert`` mesh attribute [3P, 3.x]."""
    from scipy.spatial import ConvexHull
    uniq = np.unique(np.round(verts.reshape(-1, 3), 9), axis=0)
    hull = ConvexHull(uniq)
    pts = uniq[np.sort(hull.vertices)]
    if maxhullvert <= 0 or len(pts) <= maxhullvert:
        return pts
    chosen = []
    for ax in range(3):
        for i in (int(np.argmin(pts[:, ax])), int(np.argmax(pts[:, ax]))):
            if i not in chosen:
                chosen.append(i)
    while len(chosen) < maxhullvert:
        try:
            h = ConvexHull(pts[chosen])
        except Exception:
            h = ConvexHull(pts[chosen], qhull_options="QJ")
        d = (pts @ h.equations[:, :3].T + h.equations[:, 3]).max(axis=1)
        d[chosen] = -1
        i = int(np.argmax(d))
        if d[i] < 1e-9:
            break
        chosen.append(i)
    return pts[np.sort(chosen)]


# ----------------------------------------------------------------------------- primitives
def _geom_inertia(gtype, size, density):
    """mass, inertia (3x3, geom frame, about the geom origin which is its COM)."""
    if gtype == GEOM_SPHERE:
        r = size[0]
        m = density * 4.0 / 3.0 * math.pi * r ** 3
        return m, np.eye(3) * (0.4 * m * r * r)
    if gtype == GEOM_BOX:
        x, y, z = size
        m = density * 8 * x * y * z
        return m, np.diag([m / 3 * (y * y + z * z), m / 3 * (x * x + z * z), m / 3 * (x * x + y * y)])
    if gtype == GEOM_CYLINDER:
        r, h = size[0], size[1]
        m = density * math.pi * r * r * 2 * h
        ixy = m * (3 * r * r + 4 * h * h) / 12.0
        return m, np.diag([ixy, ixy, 0.5 * m * r * r])
    if gtype == GEOM_CAPSULE:
        r, h = size[0], size[1]
        mc = density * math.pi * r * r * 2 * h
        ms = density * 4.0 / 3.0 * math.pi * r ** 3
        m = mc + ms
        izz = 0.5 * mc * r * r + 0.4 * ms * r * r
        ixy = mc * (3 * r * r + 4 * h * h) / 12.0 + ms * (0.4 * r * r + h * h + 0.75 * r * h)
        return m, np.diag([ixy, ixy, izz])
    if gtype == GEOM_PLANE:
        return 0.0, np.zeros((3, 3))
    raise ValueError(gtype)


# ----------------------------------------------------------------------------- defaults
class _Defaults:
    def __init__(self, root):
        self.classes = {}
        top = root.find("default")
        self._walk(top, {"geom": {}, "joint": {}}, "main")

    def _walk(self, el, inherited, name):
        cur = {k: dict(v) for k, v in inherited.items()}
        if el is not None:
            for kind in ("geom", "joint"):
                for d in el.findall(kind):
                    cur[kind].update(d.attrib)
        self.classes[name] = cur
        if el is not None:
            for sub in el.findall("default"):
                self._walk(sub, cur, sub.get("class"))

    def apply(self, kind, el, childclass=None):
        cls = el.get("class") or childclass or "main"
        out = dict(self.classes[cls][kind])
        out.update(el.attrib)
        return out


# ----------------------------------------------------------------------------- compiler
def _expand_includes(el, basedir):
    for i, child in enumerate(list(el)):
        if child.tag == "include":
            sub = ET.parse(os.path.join(basedir, child.get("file"))).getroot()
            _expand_includes(sub, basedir)
            idx = list(el).index(child)
            el.remove(child)
            for k, c in enumerate(list(sub)):
                el.insert(idx + k, c)
        else:
            _expand_includes(child, basedir)


def hull_planes(verts):
    """Outward face planes (n, d) with n.x + d <= 0 inside, one row per distinct face of the convex hull of ``verts``."""
    from scipy.spatial import ConvexHull
    eq = ConvexHull(verts).equations
    key = np.round(eq, 7)
    _, idx = np.unique(key, axis=0, return_index=True)
    return eq[np.sort(idx)]


def compile_mjcf(path, *, objects=None, arm_collision=False, mesh_inertia="dedup", maxhullvert=0, maxvisvert=48, maxarmhullvert=32, maxgripvisvert=32):
    """Compile an MJCF file.

    ``objects``: optional list of dicts replacing every free object of the scene (used for the
    synthetic "IT1, 4 equal boxes" configuration of SURVEY.md section 8d). Each dict has keys
    ``type`` ("box"/"sphere"), ``size``, ``pos``, ``joints`` ("slide3ball" or "free"), ``rgba``.
    """
    basedir = os.path.dirname(os.path.abspath(path))
    root = ET.parse(path).getroot()
    _expand_includes(root, basedir)
    comp = root.find("compiler")
    if comp is None or comp.get("angle", "degree") != "radian":
        raise ValueError("only angle='radian' models are supported")
    if comp.get("inertiafromgeom", "auto") != "true":
        raise ValueError("only inertiafromgeom='true' models are supported")
    meshdir = os.path.join(basedir, comp.get("meshdir", ""))
    opt = root.find("option")
    defaults = _Defaults(root)
    density = 1000.0

    meshes = {}
    for m in root.find("asset").findall("mesh"):
        meshes[m.get("name")] = os.path.join(meshdir, m.get("file"))

    bodies = [dict(name="world", parent=-1, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]))]
    joints, geoms, cams = [], [], []

    world = root.find("worldbody")
    body_elems = []

    def is_object(b):
        js = b.findall("joint") + b.findall("freejoint")
        return any(j.get("type") in ("free", "ball") or j.tag == "freejoint" for j in js)

    def add_geom(g, bid):
        a = defaults.apply("geom", g)
        gtype = _GEOM_TYPES[a.get("type", "sphere")]
        size = np.zeros(3)
        if "size" in a:
            s = _floats(a["size"])
            size[:len(s)] = s
        rgba = _floats(a["rgba"], 4) if "rgba" in a else np.array([0.5, 0.5, 0.5, 1.0])
        if "rgba" not in a and a.get("material") in _MATERIAL_RGB:
            rgba = np.array(_MATERIAL_RGB[a.get("material")])
        fr = np.array([1.0, 0.005, 0.0001])
        if "friction" in a:
            f = _floats(a["friction"])
            fr[:len(f)] = f
        solimp = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
        if "solimp" in a:
            s = _floats(a["solimp"])
            solimp[:len(s)] = s
        geoms.append(dict(name=a.get("name", ""), type=gtype, body=bid, size=size,
                          pos=_floats(a["pos"], 3) if "pos" in a else np.zeros(3),
                          quat=_orientation(g), friction=fr, condim=int(a.get("condim", 3)),
                          margin=float(a.get("margin", 0.0)),
                          solref=_floats(a["solref"], 2) if "solref" in a else np.array([0.02, 1.0]),
                          solimp=solimp, rgba=rgba, mesh=a.get("mesh"),
                          contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1))))

    def add_body(b, parent):
        bid = len(bodies)
        bodies.append(dict(name=b.get("name", f"body{bid}"), parent=parent,
                           pos=_floats(b.get("pos", "0 0 0"), 3), quat=_orientation(b)))
        for j in list(b):
            if j.tag not in ("joint", "freejoint"):
                continue
            a = defaults.apply("joint", j) if j.tag == "joint" else dict(j.attrib, type="free")
            jt = _JNT_TYPES[a.get("type", "hinge")]
            joints.append(dict(name=a.get("name", ""), type=jt, body=bid,
                               pos=_floats(a.get("pos", "0 0 0"), 3),
                               axis=_floats(a.get("axis", "0 0 1"), 3),
                               limited=a.get("limited", "false") == "true",
                               range=_floats(a.get("range", "0 0"), 2),
                               damping=float(a.get("damping", 0.0)),
                               armature=float(a.get("armature", 0.0)),
                               ref=float(a.get("ref", 0.0))))
            if float(a.get("stiffness", 0.0)) != 0.0:
                raise ValueError("joint stiffness is not supported")
        for g in b.findall("geom"):
            add_geom(g, bid)
        for c in b.findall("camera"):
            raise ValueError("body-attached cameras are not supported")
        for sub in b.findall("body"):
            add_body(sub, bid)
        return bid

    for g in world.findall("geom"):
        add_geom(g, 0)
    for c in world.findall("camera"):
        cams.append(dict(name=c.get("name"), pos=_floats(c.get("pos", "0 0 0"), 3), quat=_orientation(c),
                         fovy=float(c.get("fovy", 45.0))))
    top_bodies = world.findall("body")
    scene_objects = [b for b in top_bodies if is_object(b)]
    for b in top_bodies:
        if objects is not None and b in scene_objects:
            continue
        add_body(b, 0)
    if objects is not None:
        for k, o in enumerate(objects):
            b = ET.Element("body", name=o.get("name", f"object_{k}"),
                           pos=" ".join(repr(float(x)) for x in o["pos"]))
            if o.get("joints", "slide3ball") == "free":
                ET.SubElement(b, "joint", type="free", name=f"free_joint_{k}",
                              damping=str(o.get("damping", 0.007)))
            else:
                for ax, nm, rg in (("1 0 0", "x", "-5. 5."), ("0 1 0", "y", "-5. 5."), ("0 0 1", "z", "-2. 2.")):
                    ET.SubElement(b, "joint", type="slide", name=f"{b.get('name')}_{nm}", axis=ax, limited="true",
                                  range=rg, armature="0", damping="0")
                ET.SubElement(b, "joint", type="ball", name=f"{b.get('name')}_rot", armature="0",
                              damping=str(o.get("damping", 0.0)))
            size = o["size"] if np.ndim(o["size"]) else [o["size"]]
            ET.SubElement(b, "geom", name=b.get("name"), type=o["type"],
                          size=" ".join(repr(float(x)) for x in size),
                          rgba=" ".join(repr(float(x)) for x in o.get("rgba", (0.8, 0.6, 0.4, 1.0))))
            add_body(b, 0)

    nbody = len(bodies)
    # ---- joints / dofs layout
    qadr = vadr = 0
    for j in joints:
        j["qposadr"], j["dofadr"] = qadr, vadr
        nq_j, nv_j = {JNT_FREE: (7, 6), JNT_BALL: (4, 3), JNT_SLIDE: (1, 1), JNT_HINGE: (1, 1)}[j["type"]]
        j["nq"], j["nv"] = nq_j, nv_j
        qadr += nq_j
        vadr += nv_j
    nq, nv = qadr, vadr

    m = CompiledModel()
    m.names = dict(body=[b["name"] for b in bodies], joint=[j["name"] for j in joints],
                   geom=[g["name"] for g in geoms], camera=[c["name"] for c in cams])
    m.body_parentid = np.array([b["parent"] for b in bodies], dtype=np.int32)
    m.body_pos = np.array([b["pos"] for b in bodies])
    m.body_quat = np.array([b["quat"] for b in bodies])
    m.body_jntadr = np.full(nbody, -1, dtype=np.int32)
    m.body_jntnum = np.zeros(nbody, dtype=np.int32)
    m.body_dofadr = np.full(nbody, -1, dtype=np.int32)
    m.body_dofnum = np.zeros(nbody, dtype=np.int32)
    for k, j in enumerate(joints):
        b = j["body"]
        if m.body_jntadr[b] < 0:
            m.body_jntadr[b] = k
            m.body_dofadr[b] = j["dofadr"]
        m.body_jntnum[b] += 1
        m.body_dofnum[b] += j["nv"]
    # weld groups / trees
    m.body_weldid = np.zeros(nbody, dtype=np.int32)
    for b in range(1, nbody):
        m.body_weldid[b] = b if m.body_jntnum[b] > 0 else m.body_weldid[m.body_parentid[b]]
    m.body_treeid = np.full(nbody, -1, dtype=np.int32)
    ntree = 0
    for b in range(1, nbody):
        p = m.body_parentid[b]
        if m.body_treeid[p] >= 0:
            m.body_treeid[b] = m.body_treeid[p]
        elif m.body_jntnum[b] > 0:
            m.body_treeid[b] = ntree
            ntree += 1
    m.jnt_type = np.array([j["type"] for j in joints], dtype=np.int32)
    m.jnt_qposadr = np.array([j["qposadr"] for j in joints], dtype=np.int32)
    m.jnt_dofadr = np.array([j["dofadr"] for j in joints], dtype=np.int32)
    m.jnt_bodyid = np.array([j["body"] for j in joints], dtype=np.int32)
    m.jnt_pos = np.array([j["pos"] for j in joints])
    m.jnt_axis = np.array([j["axis"] / np.linalg.norm(j["axis"]) for j in joints])
    m.jnt_limited = np.array([int(j["limited"] and j["type"] in (JNT_SLIDE, JNT_HINGE)) for j in joints], dtype=np.int32)
    m.jnt_range = np.array([j["range"] for j in joints])
    m.qpos0 = np.zeros(nq)
    m.dof_bodyid = np.zeros(nv, dtype=np.int32)
    m.dof_jntid = np.zeros(nv, dtype=np.int32)
    m.dof_parentid = np.full(nv, -1, dtype=np.int32)
    m.dof_armature = np.zeros(nv)
    m.dof_damping = np.zeros(nv)
    m.dof_treeid = np.zeros(nv, dtype=np.int32)
    for k, j in enumerate(joints):
        b = j["body"]
        if j["type"] == JNT_FREE:
            m.qpos0[j["qposadr"]:j["qposadr"] + 3] = bodies[b]["pos"]
            m.qpos0[j["qposadr"] + 3:j["qposadr"] + 7] = bodies[b]["quat"]
        elif j["type"] == JNT_BALL:
            m.qpos0[j["qposadr"]:j["qposadr"] + 4] = [1, 0, 0, 0]
        else:
            m.qpos0[j["qposadr"]] = j["ref"]
        for d in range(j["nv"]):
            i = j["dofadr"] + d
            m.dof_bodyid[i] = b
            m.dof_jntid[i] = k
            m.dof_armature[i] = j["armature"]
            m.dof_damping[i] = j["damping"]
            m.dof_treeid[i] = m.body_treeid[b]
    last_dof_of_body = {}
    for i in range(nv):
        b = m.dof_bodyid[i]
        if i > 0 and m.dof_bodyid[i - 1] == b:
            m.dof_parentid[i] = i - 1
        else:
            p = m.body_parentid[b]
            while p > 0 and p not in last_dof_of_body:
                p = m.body_parentid[p]
            m.dof_parentid[i] = last_dof_of_body.get(p, -1)
        last_dof_of_body[b] = i
    m.tree_dofadr = np.array([int(np.nonzero(m.dof_treeid == t)[0][0]) for t in range(ntree)], dtype=np.int32)
    m.tree_dofnum = np.array([int((m.dof_treeid == t).sum()) for t in range(ntree)], dtype=np.int32)

    # ---- geoms, meshes
    mesh_cache, mesh_vert, mesh_adr, mesh_num, mesh_names = {}, [], [], [], []
    vis_adr, vis_num, vis_plane = [], [], []
    ngeom = len(geoms)
    m.geom_type = np.array([g["type"] for g in geoms], dtype=np.int32)
    m.geom_bodyid = np.array([g["body"] for g in geoms], dtype=np.int32)
    m.geom_size = np.array([g["size"] for g in geoms])
    m.geom_pos = np.array([g["pos"] for g in geoms])
    m.geom_quat = np.array([g["quat"] for g in geoms])
    m.geom_friction = np.array([g["friction"] for g in geoms])
    m.geom_condim = np.array([g["condim"] for g in geoms], dtype=np.int32)
    m.geom_margin = np.array([g["margin"] for g in geoms])
    m.geom_solref = np.array([g["solref"] for g in geoms])
    m.geom_solimp = np.array([g["solimp"] for g in geoms])
    m.geom_rgba = np.array([g["rgba"] for g in geoms])
    m.geom_meshid = np.full(ngeom, -1, dtype=np.int32)
    m.geom_rbound = np.zeros(ngeom)
    m.geom_collide = np.ones(ngeom, dtype=np.int32)
    geom_mass = np.zeros(ngeom)
    geom_com = np.zeros((ngeom, 3))         # in geom frame
    geom_inertia = np.zeros((ngeom, 3, 3))  # geom frame, about com
    for k, g in enumerate(geoms):
        if g["type"] == GEOM_MESH:
            name = g["mesh"]
            if name not in mesh_cache:
                tris = load_stl(meshes[name])
                # gripper meshes collide with their full hulls (maxhullvert = 0). The seven arm-link hulls (1.7k-2.8k vertices; they only ever touch an
                # object when the arm crashes into the bin) are capped at maxarmhullvert when they collide at all.
                if name in _ARM_MESHES:
                    hull = convex_hull_vertices(tris, maxarmhullvert if arm_collision else 0)
                else:
                    hull = convex_hull_vertices(tris, maxhullvert)
                if mesh_inertia == "legacy":
                    mi = mesh_inertia_legacy(tris, density)
                elif mesh_inertia == "signed":
                    mi = mesh_inertia_signed(tris, density)
                elif mesh_inertia == "dedup":
                    mi = mesh_inertia_signed(tris, density, dedup=True)
                else:
                    raise ValueError(mesh_inertia)
                mesh_cache[name] = (len(mesh_adr), mi)
                mesh_adr.append(sum(mesh_num))
                mesh_num.append(len(hull))
                mesh_vert.append(hull)
                mesh_names.append(name)
                # faces of the (capped) hull for the ray caster: the collision hull where one exists, else <= maxvisvert vertices
                vis = convex_hull_vertices(tris, maxgripvisvert if name not in _ARM_MESHES else maxvisvert)
                pl = hull_planes(vis)
                vis_adr.append(sum(vis_num)); vis_num.append(len(pl)); vis_plane.append(pl)
            mid, (mass, com, inert) = mesh_cache[name]
            m.geom_meshid[k] = mid
            geom_mass[k], geom_com[k], geom_inertia[k] = mass, com, inert
            m.geom_rbound[k] = np.linalg.norm(mesh_vert[mid], axis=1).max()
            m.geom_size[k] = np.abs(mesh_vert[mid]).max(0)
            if name in _ARM_MESHES and not arm_collision:
                m.geom_collide[k] = 0
        else:
            geom_mass[k], geom_inertia[k] = _geom_inertia(g["type"], g["size"], density)
            s = g["size"]
            m.geom_rbound[k] = {GEOM_SPHERE: s[0], GEOM_BOX: np.linalg.norm(s), GEOM_CAPSULE: s[0] + s[1],
                                GEOM_CYLINDER: math.hypot(s[0], s[1]), GEOM_PLANE: 0.0}[g["type"]]
    m.names["mesh"] = mesh_names
    m.mesh_vertadr = np.array(mesh_adr, dtype=np.int32)
    m.mesh_vertnum = np.array(mesh_num, dtype=np.int32)
    m.mesh_vert = np.concatenate(mesh_vert) if mesh_vert else np.zeros((0, 3))
    m.vis_planeadr = np.array(vis_adr, dtype=np.int32)
    m.vis_planenum = np.array(vis_num, dtype=np.int32)
    m.vis_plane = np.concatenate(vis_plane) if vis_plane else np.zeros((0, 4))

    # ---- body inertias from geoms (inertiafromgeom="true")
    m.body_mass = np.zeros(nbody)
    m.body_ipos = np.zeros((nbody, 3))
    m.body_inertia = np.zeros((nbody, 6))    # xx yy zz xy xz yz, body frame, about ipos
    for b in range(nbody):
        gs = [k for k in range(ngeom) if geoms[k]["body"] == b and geom_mass[k] > 0]
        if not gs:
            continue
        mass = sum(geom_mass[k] for k in gs)
        com = sum(geom_mass[k] * (geoms[k]["pos"] + quat_to_mat(geoms[k]["quat"]) @ geom_com[k]) for k in gs) / mass
        I = np.zeros((3, 3))
        for k in gs:
            R = quat_to_mat(geoms[k]["quat"])
            c = geoms[k]["pos"] + R @ geom_com[k] - com
            I += R @ geom_inertia[k] @ R.T + geom_mass[k] * (c.dot(c) * np.eye(3) - np.outer(c, c))
        m.body_mass[b] = mass
        m.body_ipos[b] = com
        m.body_inertia[b] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]

    # ---- collision pair list
    excl = set()
    con = root.find("contact")
    if con is not None:
        for e in con.findall("exclude"):
            b1, b2 = m.names["body"].index(e.get("body1")), m.names["body"].index(e.get("body2"))
            excl.add((min(b1, b2), max(b1, b2)))
    weld = m.body_weldid
    pairs = []
    for g1 in range(ngeom):
        for g2 in range(g1 + 1, ngeom):
            b1, b2 = geoms[g1]["body"], geoms[g2]["body"]
            if not (m.geom_collide[g1] and m.geom_collide[g2]):
                continue
            if b1 == b2 or (weld[b1] == 0 and weld[b2] == 0) or weld[b1] == weld[b2]:
                continue
            w1, w2 = weld[b1], weld[b2]
            p1, p2 = weld[m.body_parentid[w1]] if w1 else -1, weld[m.body_parentid[w2]] if w2 else -1
            if w1 != 0 and w2 != 0 and (w1 == p2 or w2 == p1):
                continue
            if (min(b1, b2), max(b1, b2)) in excl:
                continue
            if not ((geoms[g1]["contype"] & geoms[g2]["conaffinity"]) or (geoms[g2]["contype"] & geoms[g1]["conaffinity"])):
                continue
            if geoms[g1]["type"] == GEOM_PLANE and geoms[g2]["type"] == GEOM_PLANE:
                continue
            pairs.append((g1, g2))
    m.pair_geom1 = np.array([p[0] for p in pairs], dtype=np.int32)
    m.pair_geom2 = np.array([p[1] for p in pairs], dtype=np.int32)

    # ---- equality (joint) constraints
    eqs = []
    eq = root.find("equality")
    if eq is not None:
        for e in eq:
            if e.tag != "joint":
                raise ValueError(f"equality type {e.tag} is not supported")
            pc = np.zeros(5)
            p = _floats(e.get("polycoef", "0 1 0 0 0"))
            pc[:len(p)] = p
            eqs.append((m.names["joint"].index(e.get("joint1")), m.names["joint"].index(e.get("joint2")), pc,
                        _floats(e.get("solref", "0.02 1"), 2),
                        _floats(e.get("solimp", "0.9 0.95 0.001 0.5 2"))))
    m.eq_jnt1 = np.array([e[0] for e in eqs], dtype=np.int32)
    m.eq_jnt2 = np.array([e[1] for e in eqs], dtype=np.int32)
    m.eq_polycoef = np.array([e[2] for e in eqs]).reshape(-1, 5)
    m.eq_solref = np.array([e[3] for e in eqs]).reshape(-1, 2)
    m.eq_solimp = np.array([np.concatenate([e[4], [0.5, 2.0]])[:5] for e in eqs]).reshape(-1, 5)

    # ---- actuators (motors only)
    acts = []
    for a in root.find("actuator"):
        if a.tag != "motor":
            raise ValueError(f"actuator type {a.tag} is not supported")
        acts.append((a.get("name"), m.names["joint"].index(a.get("joint")), float(a.get("gear", "1").split()[0]),
                     _floats(a.get("ctrlrange", "0 0"), 2), a.get("ctrllimited", "false") == "true"))
    m.names["actuator"] = [a[0] for a in acts]
    m.act_jntid = np.array([a[1] for a in acts], dtype=np.int32)
    m.act_gear = np.array([a[2] for a in acts])
    m.act_ctrlrange = np.array([a[3] for a in acts])
    m.act_ctrllimited = np.array([int(a[4]) for a in acts], dtype=np.int32)

    # ---- options, cameras, visual
    vis = root.find("visual")
    vmap = vis.find("map") if vis is not None else None
    m.opt = dict(timestep=float(opt.get("timestep", 0.002)), iterations=int(opt.get("iterations", 100)),
                 tolerance=float(opt.get("tolerance", 1e-8)), impratio=float(opt.get("impratio", 1.0)),
                 gravity=[0.0, 0.0, -9.81], jnt_solref=[0.02, 1.0], jnt_solimp=[0.9, 0.95, 0.001, 0.5, 2.0],
                 znear=float(vmap.get("znear", 0.01)) if vmap is not None else 0.01,
                 zfar=float(vmap.get("zfar", 50.0)) if vmap is not None else 50.0)
    m.cam_pos = np.array([c["pos"] for c in cams]).reshape(-1, 3)
    m.cam_mat = np.array([quat_to_mat(c["quat"]).reshape(9) for c in cams]).reshape(-1, 9)
    m.cam_fovy = np.array([c["fovy"] for c in cams])

    from .refdyn import finalize_model
    finalize_model(m)
    return m


# flat albedo per material (textures are out of scope)
_MATERIAL_RGB = {"ur5_mat": (0.45, 0.45, 0.45, 1.0), "gripper_mat": (0.45, 0.45, 0.45, 1.0),
                 "floor_mat": (0.15, 0.25, 0.35, 1.0), "geom": (0.8, 0.6, 0.4, 1.0),
                 "bench_mat": (0.6, 0.6, 0.62, 1.0), "tablecube": (0.72, 0.52, 0.32, 1.0)}
