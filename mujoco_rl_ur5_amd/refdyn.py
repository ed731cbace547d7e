"""Model-compile-time rigid-body helpers in numpy (forward kinematics, Jacobians, mass matrix).

Used to derive the constants MuJoCo's compiler would produce at ``qpos0`` [3P, SURVEY.md C.4]:
``body_invweight0``, ``dof_invweight0``, ``stat.meaninertia``, ``stat.extent``. The mass matrix here is
assembled from per-body Jacobians (sum_b J_b^T diag(m, I) J_b) -- deliberately a different
algorithm from the CRBA used by the oracle and the HIP engine, so tests can cross-check them.
Not on the hot path.
"""
from __future__ import annotations

import numpy as np

from .model import JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE, GEOM_PLANE


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def _qmat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def _axisangle(axis, ang):
    s = np.sin(ang / 2)
    return np.array([np.cos(ang / 2), axis[0] * s, axis[1] * s, axis[2] * s])


def forward_kinematics(m, qpos):
    """Returns dict(xpos, xmat, xquat, xipos, xanchor, xaxis) following MuJoCo's joint conventions (C.1)."""
    nb = m.nbody
    xpos, xquat = np.zeros((nb, 3)), np.zeros((nb, 4))
    xquat[0] = [1, 0, 0, 0]
    nj = len(m.jnt_type)
    xanchor, xaxis = np.zeros((nj, 3)), np.zeros((nj, 3))
    for b in range(1, nb):
        p = m.body_parentid[b]
        R = _qmat(xquat[p])
        pos = xpos[p] + R @ m.body_pos[b]
        quat = _qmul(xquat[p], m.body_quat[b])
        for j in range(m.body_jntadr[b], m.body_jntadr[b] + m.body_jntnum[b]):
            qa = m.jnt_qposadr[j]
            t = m.jnt_type[j]
            if t == JNT_FREE:
                pos = np.array(qpos[qa:qa + 3], dtype=float)
                quat = np.array(qpos[qa + 3:qa + 7], dtype=float)
                quat /= np.linalg.norm(quat)
                xanchor[j] = pos
                xaxis[j] = [0, 0, 1]
                continue
            Rb = _qmat(quat)
            xanchor[j] = pos + Rb @ m.jnt_pos[j]
            xaxis[j] = Rb @ m.jnt_axis[j]
            if t == JNT_SLIDE:
                pos = pos + xaxis[j] * (qpos[qa] - m.qpos0[qa])
            elif t == JNT_HINGE:
                quat = _qmul(quat, _axisangle(m.jnt_axis[j], qpos[qa] - m.qpos0[qa]))
                pos = xanchor[j] - _qmat(quat) @ m.jnt_pos[j]
            elif t == JNT_BALL:
                q = np.array(qpos[qa:qa + 4], dtype=float)
                quat = _qmul(quat, q / np.linalg.norm(q))
                pos = xanchor[j] - _qmat(quat) @ m.jnt_pos[j]
        xpos[b], xquat[b] = pos, quat / np.linalg.norm(quat)
    xmat = np.array([_qmat(q) for q in xquat])
    xipos = xpos + np.einsum("bij,bj->bi", xmat, m.body_ipos)
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, xanchor=xanchor, xaxis=xaxis)


def body_jacobian(m, fk, body, point):
    """(jacp, jacr): 3 x nv translational Jacobian of `point` attached to `body`, and rotational Jacobian."""
    nv = m.nv
    jacp, jacr = np.zeros((3, nv)), np.zeros((3, nv))
    b = body
    while b > 0:
        for j in range(m.body_jntadr[b], m.body_jntadr[b] + m.body_jntnum[b]):
            d, t = m.jnt_dofadr[j], m.jnt_type[j]
            if t == JNT_SLIDE:
                jacp[:, d] = fk["xaxis"][j]
            elif t == JNT_HINGE:
                jacr[:, d] = fk["xaxis"][j]
                jacp[:, d] = np.cross(fk["xaxis"][j], point - fk["xanchor"][j])
            else:
                if t == JNT_FREE:
                    jacp[:, d:d + 3] = np.eye(3)
                    d += 3
                R = fk["xmat"][b]
                for k in range(3):
                    jacr[:, d + k] = R[:, k]
                    jacp[:, d + k] = np.cross(R[:, k], point - fk["xanchor"][j])
        b = m.body_parentid[b]
    return jacp, jacr


def _inertia_mat(v6):
    xx, yy, zz, xy, xz, yz = v6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


def mass_matrix(m, qpos):
    fk = forward_kinematics(m, qpos)
    M = np.diag(np.asarray(m.dof_armature, dtype=float))
    for b in range(1, m.nbody):
        if m.body_mass[b] <= 0:
            continue
        jp, jr = body_jacobian(m, fk, b, fk["xipos"][b])
        Iw = fk["xmat"][b] @ _inertia_mat(m.body_inertia[b]) @ fk["xmat"][b].T
        M += m.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
    return M, fk


def finalize_model(m):
    """Fill body_invweight0, dof_invweight0, opt['meaninertia'], opt['extent'] at qpos0."""
    nv = m.nv
    M, fk = mass_matrix(m, m.qpos0)
    Minv = np.linalg.inv(M) if nv else np.zeros((0, 0))
    m.body_invweight0 = np.zeros((m.nbody, 2))
    for b in range(1, m.nbody):
        if m.body_weldid[b] == 0:
            continue
        jp, jr = body_jacobian(m, fk, b, fk["xipos"][b])
        m.body_invweight0[b, 0] = np.trace(jp @ Minv @ jp.T) / 3.0
        m.body_invweight0[b, 1] = np.trace(jr @ Minv @ jr.T) / 3.0
    m.dof_invweight0 = np.zeros(nv)
    d = np.diag(Minv)
    for j in range(len(m.jnt_type)):
        a, t = m.jnt_dofadr[j], m.jnt_type[j]
        if t in (JNT_SLIDE, JNT_HINGE):
            m.dof_invweight0[a] = d[a]
        elif t == JNT_BALL:
            m.dof_invweight0[a:a + 3] = d[a:a + 3].mean()
        else:
            m.dof_invweight0[a:a + 3] = d[a:a + 3].mean()
            m.dof_invweight0[a + 3:a + 6] = d[a + 3:a + 6].mean()
    m.opt["meaninertia"] = float(np.diag(M).mean()) if nv else 1.0
    # stat.extent: half the largest side of the bounding box of all (bounded) geoms at qpos0 [3P, C.7]
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for g in range(m.ngeom):
        if m.geom_type[g] == GEOM_PLANE:
            continue
        b = m.geom_bodyid[g]
        c = fk["xpos"][b] + fk["xmat"][b] @ m.geom_pos[g]
        lo, hi = np.minimum(lo, c - m.geom_rbound[g]), np.maximum(hi, c + m.geom_rbound[g])
    m.opt["extent"] = float(max(0.5 * (hi - lo).max(), 1e-6))
    m._pack_opt()
