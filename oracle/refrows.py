"""Constraint rows of one forward pass, restated in numpy -- TEST INFRASTRUCTURE ONLY (like everything under oracle/).

What `mj_makeConstraint` / `mj_makeImpedance` / `mj_referenceConstraint` [3P: MuJoCo, "Computation" chapter -- solref / solimp -> stiffness, damping, impedance;
pyramidal friction cones; regularisation from the inverse weights at qpos0; SURVEY.md Appendix C.4] build for the scenes of the reference
(`UR5gripper_2_finger.xml:19-22,25-38,333`: tolerance / impratio, geom defaults solref ".01 1" solimp ".99 .99 .01", the `fingers` joint equality;
`UR5gripper_2_finger_many_objects.xml:28`: condim 6): the dense Jacobian row, residual, margin, reference acceleration `aref` and regulariser `R` of
  * the joint equality  q1 - q1_0 = poly(q2 - q2_0),
  * violated joint limits,
  * contacts with condim 1 / 3 / 4 / 6 as pyramids of 2 (condim - 1) rows  J_n +- mu_k J_k.

It shares NO code with oracle/ur5_oracle.cpp or the HIP engine: kinematics and Jacobians come from `mujoco_rl_ur5_amd/refdyn.py` (Jacobian sums in numpy),
the rows are dense numpy vectors, nothing is in "twist space". tests/test_constraint_rows.py checks the oracle's rows against it value by value, and checks
that the accelerations the oracle AND the engine return are the minimiser of the convex problem these rows define (zero gradient) -- which pins the engine's
fused row construction + Newton solve against an independent text of the published formulas.
"""
from __future__ import annotations

import numpy as np

from mujoco_rl_ur5_amd.model import JNT_HINGE, JNT_SLIDE
from mujoco_rl_ur5_amd.refdyn import body_jacobian, forward_kinematics, mass_matrix

MINVAL = 1e-15


def impedance(solimp, x_abs):
    """d(r) of the solimp sigmoid: dmin .. dmax over `width`, midpoint / power shape."""
    dmin, dmax = np.clip(solimp[0], 0.0001, 0.9999), np.clip(solimp[1], 0.0001, 0.9999)
    width, mid, power = solimp[2], solimp[3], solimp[4]
    if dmin == dmax or width <= MINVAL:
        return 0.5 * (dmin + dmax)
    x = x_abs / width
    if x >= 1:
        return dmax
    if x <= 0:
        return dmin
    if power == 1:
        y = x
    elif x <= mid:
        y = (x / mid) ** power * mid
    else:
        y = 1 - ((1 - x) / (1 - mid)) ** power * (1 - mid)
    return dmin + y * (dmax - dmin)


def stiffness_damping(solref, solimp, timestep):
    """(k, b) of a positive solref = (time constant, damping ratio); the time constant is kept >= 2 h (refsafe)."""
    tc, dr = max(solref[0], 2 * timestep), solref[1]
    dmax = np.clip(solimp[1], 0.0001, 0.9999)
    return 1.0 / (dmax * dmax * tc * tc * dr * dr), 2.0 / (dmax * tc)


def contact_frame(normal):
    """x = normal, y = the world y (or z) axis made orthogonal to it, z = x cross y (mju_makeFrame [3P])."""
    x = np.asarray(normal, dtype=float)
    y = np.array([0.0, 1.0, 0.0]) if abs(x[1]) < 0.5 else np.array([0.0, 0.0, 1.0])
    y = y - x * (x @ y)
    y /= np.linalg.norm(y)
    return np.stack([x, y, np.cross(x, y)])


class Rows:
    def __init__(self, nv):
        self.J, self.pos, self.margin, self.aref, self.R, self.unilateral, self.kind = [], [], [], [], [], [], []
        self.nv = nv

    def add(self, J, pos, margin, diag_approx, solref, solimp, qvel, timestep, unilateral, kind):
        imp = impedance(solimp, abs(pos - margin))
        k, b = stiffness_damping(solref, solimp, timestep)
        self.J.append(J)
        self.pos.append(pos)
        self.margin.append(margin)
        self.aref.append(-b * (J @ qvel) - k * imp * (pos - margin))
        self.R.append(max(MINVAL, (1 - imp) * diag_approx / imp))
        self.unilateral.append(unilateral)
        self.kind.append(kind)

    def finish(self):
        for k in ("J", "pos", "margin", "aref", "R"):
            setattr(self, k, np.array(getattr(self, k), dtype=float).reshape((len(self.pos), self.nv) if k == "J" else (len(self.pos),)))
        self.unilateral = np.array(self.unilateral, dtype=bool)
        return self


def build_rows(m, qpos, qvel, contacts):
    """contacts: rows of (dist, px, py, pz, nx, ny, nz, geom1, geom2, ...) -- the contact GEOMETRY is an input (it has its own closed-form tests)."""
    qpos, qvel = np.asarray(qpos, dtype=float), np.asarray(qvel, dtype=float)
    h, nv = m.opt["timestep"], m.nv
    fk = forward_kinematics(m, qpos)
    rows = Rows(nv)
    for e in range(len(m.eq_jnt1)):                                           # joint equality (UR5gripper_2_finger.xml:333)
        j1, j2 = int(m.eq_jnt1[e]), int(m.eq_jnt2[e])
        q1, q2, d1, d2 = m.jnt_qposadr[j1], m.jnt_qposadr[j2], m.jnt_dofadr[j1], m.jnt_dofadr[j2]
        c = m.eq_polycoef[e]
        x = qpos[q2] - m.qpos0[q2]
        J = np.zeros(nv)
        J[d1], J[d2] = 1.0, -np.polyval(np.polyder(c[::-1]), x)
        rows.add(J, (qpos[q1] - m.qpos0[q1]) - np.polyval(c[::-1], x), 0.0, m.dof_invweight0[d1] + m.dof_invweight0[d2], m.eq_solref[e], m.eq_solimp[e],
                 qvel, h, False, "equality")
    for j in range(len(m.jnt_type)):                                          # joint limits: a row per violated side
        if not m.jnt_limited[j] or m.jnt_type[j] not in (JNT_HINGE, JNT_SLIDE):
            continue
        qa, d = m.jnt_qposadr[j], m.jnt_dofadr[j]
        for side, dist in ((1.0, qpos[qa] - m.jnt_range[j][0]), (-1.0, m.jnt_range[j][1] - qpos[qa])):
            if dist < 0:
                J = np.zeros(nv)
                J[d] = side
                rows.add(J, dist, 0.0, m.dof_invweight0[d], m.opt["jnt_solref"], m.opt["jnt_solimp"], qvel, h, True, "limit")
    for c in np.atleast_2d(np.asarray(contacts, dtype=float)) if len(contacts) else []:
        dist, pos, normal, g1, g2 = c[0], c[1:4], c[4:7], int(c[7]), int(c[8])
        b1, b2 = int(m.geom_bodyid[g1]), int(m.geom_bodyid[g2])
        dim = int(max(m.geom_condim[g1], m.geom_condim[g2]))
        fri3 = np.maximum(m.geom_friction[g1], m.geom_friction[g2])           # (slide, spin, roll)
        mu = [fri3[0], fri3[0], fri3[1], fri3[2], fri3[2]]                    # per friction direction: 2 tangents, torsion, 2 rolling
        margin = max(m.geom_margin[g1], m.geom_margin[g2])
        solref, solimp = 0.5 * (m.geom_solref[g1] + m.geom_solref[g2]), 0.5 * (m.geom_solimp[g1] + m.geom_solimp[g2])   # solmix 1 : 1
        F = contact_frame(normal)
        jp1, jr1 = body_jacobian(m, fk, b1, pos)
        jp2, jr2 = body_jacobian(m, fk, b2, pos)
        base = [F[k] @ (jp2 - jp1) for k in range(3)] + [F[k] @ (jr2 - jr1) for k in range(3)]   # relative velocity of body 2 w.r.t. body 1 in the contact frame
        tran = m.body_invweight0[b1][0] + m.body_invweight0[b2][0]
        first = len(rows.pos)
        if dim == 1:
            rows.add(base[0], dist, margin, tran, solref, solimp, qvel, h, True, "contact")
            continue
        for k in range(1, dim):
            for sgn in (1.0, -1.0):
                rows.add(base[0] + sgn * mu[k - 1] * base[k], dist, margin, tran + mu[0] * mu[0] * tran, solref, solimp, qvel, h, True, "contact")
        mu_reg = mu[0] * np.sqrt(1.0 / max(MINVAL, m.opt["impratio"]))       # pyramidal cone: every row of the contact gets R = 2 mu~^2 R[first row]
        r_py = 2 * mu_reg * mu_reg * rows.R[first]
        for i in range(first, len(rows.pos)):
            rows.R[i] = r_py
    return rows.finish()


def primal_gradient(m, rows, qpos, qfrc_smooth, qacc):
    """Gradient of  1/2 (a - a_smooth)' M (a - a_smooth) + sum_i s_i(J_i a - aref_i),  s_i(r) = r^2 / (2 R_i) on active rows (bilateral, or r < 0),
    at a = qacc, with M a_smooth = qfrc_smooth. Returns (gradient, scale): scale = |M a| + |qfrc_smooth| + |J' f| for a relative test."""
    M, _ = mass_matrix(m, np.asarray(qpos, dtype=float))
    qacc = np.asarray(qacc, dtype=float)
    g = M @ qacc - np.asarray(qfrc_smooth, dtype=float)
    scale = np.abs(g).max() + np.abs(qfrc_smooth).max()
    if len(rows.pos):
        jar = rows.J @ qacc - rows.aref
        f = np.where(~rows.unilateral | (jar < 0), -jar / rows.R, 0.0)
        g = g - rows.J.T @ f
        scale += np.abs(rows.J.T @ f).max()
    return g, scale
