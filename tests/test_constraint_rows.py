"""Independent pin of the constraint rows (round-3 verdict, item 3): oracle/refrows.py restates solref / solimp -> (k, b, impedance) -> aref, R and the
pyramid rows in dense numpy (Jacobians by refdyn.py's Jacobian sums) -- nothing shared with the oracle's C++ or the engine's twist-space code. Checked here:
  (1) every row of the oracle (pos, aref, R) equals the numpy row to 1e-10, on grasp states with up to ~20 contacts (condim 4), settled piles (condim 6:
      10 rows per contact), a violated joint limit and the `fingers` equality (UR5gripper_2_finger.xml:19-22,25-38,333);
  (2) the constrained acceleration the oracle returns is the minimiser of the convex problem those numpy rows define: its gradient vanishes;
  (3) the same for the acceleration the ENGINE returns (lane emulation here, wavefront emulation, the MI355X with -m gpu) with rows rebuilt from the engine's
      own contact list -- the engine never forms rows (it works on body twists / wrenches), so this is the only place its fused construction meets a
      row-by-row text of the formulas."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from oracle import refrows
from oracle.oracle import Oracle

MANY = "/UR5+gripper/UR5gripper_2_finger_many_objects.xml"
TWOF = "/UR5+gripper/UR5gripper_2_finger.xml"


def _grasp_states(spec, seeds):
    from solver_crosscheck import grasp_states                                  # the states of profiles/r03_solver_crosscheck.json: closed / lifting / carrying
    out = []
    for s in seeds:
        m, sts = grasp_states(spec, s, (s % 2) * 3)
        out += [(m, f"{spec}:{s}:{tag}", qpos, qvel, warm, pid, ctrl) for tag, qpos, qvel, warm, pid, ctrl in sts]
    return out


def _limit_state(spec):
    """The gripper driven past its joint range and the shoulder-lift joint past its own: limit rows on both, the equality row far from rest."""
    m = load_model(spec)
    o = Oracle(m)
    o.reset(20, 1, True)
    s = o.get_state()
    q, v = s["qpos"].copy(), s["qvel"].copy()
    lim = [j for j in range(len(m.jnt_type)) if m.jnt_limited[j] and m.jnt_dofadr[j] < 8]
    assert len(lim) >= 2
    for j in lim[:2]:
        q[m.jnt_qposadr[j]] = m.jnt_range[j][1] + 0.03
        v[m.jnt_dofadr[j]] = 0.4
    for j in lim[2:3]:
        q[m.jnt_qposadr[j]] = m.jnt_range[j][0] - 0.02
    return (m, f"{spec}:limits", q, v, s["warmstart"], s["pid"], o.get_ctrl())


def _pile_states():
    m = load_model(MANY)
    out = []
    for seed in (31, 32):
        o = Oracle(m)
        o.reset(seed, 1, False)
        o.step(260)                                                              # the pile has landed: dozens of condim-6 contacts, objects still moving
        s = o.get_state()
        out.append((m, f"pile:{seed}", s["qpos"], s["qvel"], s["warmstart"], s["pid"], o.get_ctrl()))
    return out


@pytest.fixture(scope="module")
def states():
    return _grasp_states("it1_4box", range(20, 32)) + _grasp_states(TWOF, range(20, 32)) + [_limit_state("it1_4box"), _limit_state(TWOF)] + _pile_states()


def _oracle_at(m, qpos, qvel, warm, pid, ctrl):
    o = Oracle(m)
    o.set_state(qpos=qpos, qvel=qvel, warmstart=warm, pid=pid)
    o.set_ctrl(ctrl)
    o.forward()
    return o


def test_oracle_rows_equal_the_numpy_restatement_and_its_solution_is_the_minimiser(states):
    kinds, worst_row, worst_grad, nrows, ncon6 = set(), 0.0, 0.0, 0, 0
    for m, tag, qpos, qvel, warm, pid, ctrl in states:
        o = _oracle_at(m, qpos, qvel, warm, pid, ctrl)
        con, ro = o.contacts(), o.rows()
        R = refrows.build_rows(m, qpos, qvel, con)
        assert len(R.pos) == len(ro), (tag, len(R.pos), len(ro))
        kinds |= set(R.kind)
        nrows += len(ro)
        ncon6 += int(sum(c[9] == 6 for c in con))
        for i in range(len(ro)):
            pos, aref, Rr, _, _, uni = ro[i]
            assert abs(pos - R.pos[i]) <= 1e-12 and bool(uni) == bool(R.unilateral[i]), (tag, i)
            e_aref, e_R = abs(aref - R.aref[i]) / max(1.0, abs(aref)), abs(Rr - R.R[i]) / Rr
            worst_row = max(worst_row, e_aref, e_R)
            assert e_aref < 1e-10 and e_R < 1e-10, (tag, i, R.kind[i], aref, R.aref[i], Rr, R.R[i])
        # the oracle's solution against the numpy problem: M (numpy, Jacobian sums) a - qfrc_smooth - J' f(J a - aref) = 0
        fs = o.vec("qfrc_passive") - o.vec("qfrc_bias") + o.vec("qfrc_actuator")
        g, scale = refrows.primal_gradient(m, R, qpos, fs, o.vec("qacc"))
        worst_grad = max(worst_grad, np.abs(g).max() / scale)
        assert np.abs(g).max() < 1e-8 * scale, (tag, np.abs(g).max(), scale)
    assert kinds == {"equality", "limit", "contact"} and nrows > 4000 and ncon6 > 40, (kinds, nrows, ncon6)
    print(f"rows {nrows}, worst relative row error {worst_row:.1e}, worst relative gradient {worst_grad:.1e}")


def _engine_solution_is_the_minimiser(states, lib_path, tol, only=None):
    worst = 0.0
    for m, tag, qpos, qvel, warm, pid, ctrl in states:
        if only is not None and not any(k in tag for k in only):
            continue
        sim = BatchSim(m, 1, lib_path=lib_path)
        sim.set_state(qpos=qpos[None], qvel=qvel[None], warmstart=warm[None], pid=pid[None])
        sim.set_ctrl(ctrl[None])
        d = sim.forward_debug()
        n = int(d["ncon"][0])
        con = d["contacts"][0][:n].copy()                                        # the ENGINE's contacts: dist, point, normal, geom1, geom2
        collidable = np.flatnonzero(np.asarray(m.geom_collide) != 0)             # the engine numbers the collidable geoms only (ur5sim_host.h build_model)
        con[:, 7], con[:, 8] = collidable[con[:, 7].astype(int)], collidable[con[:, 8].astype(int)]
        R = refrows.build_rows(m, qpos, qvel, con)
        g, scale = refrows.primal_gradient(m, R, qpos, d["qfrc_smooth"][0][:m.nv], d["qacc"][0][:m.nv])
        worst = max(worst, np.abs(g).max() / scale)
        assert np.abs(g).max() < tol * scale, (tag, n, np.abs(g).max(), scale)
        assert sim.counters()["status"][0] == 0
    return worst


def test_engine_solution_minimises_the_numpy_problem_lane_emulation(states, emul_lib):
    _engine_solution_is_the_minimiser(states, emul_lib, 1e-8)


def test_engine_solution_minimises_the_numpy_problem_wavefront_emulation(states, simt_lib):
    _engine_solution_is_the_minimiser(states, simt_lib, 1e-8, only=("it1_4box:20", "2_finger.xml:21", "limits", "pile:31"))


@pytest.mark.gpu
def test_engine_solution_minimises_the_numpy_problem_on_gpu(states):
    worst = _engine_solution_is_the_minimiser(states, None, 1e-8)
    print(f"worst relative gradient on the MI355X: {worst:.1e}")
