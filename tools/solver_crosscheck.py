#!/usr/bin/env python3
"""Oracle-side solver cross-check on multi-contact GRASP states: Newton (the solver the reference runs, MuJoCo's default [3P]) against projected
Gauss-Seidel on the same constraint rows with 100 (MuJoCo's cap), 1e4 and 1e6 sweeps. The contact problem is a strictly convex QP
(UR5gripper_2_finger.xml:19-22: soft constraints, R > 0 on every row): its minimiser is unique and is the only point with zero gradient. The check
therefore reports (a) the gradient of the primal objective at Newton's solution -- zero to rounding: Newton IS at the optimum --, (b) that PGS,
which shares only the row construction with Newton, never gets below Newton's cost and closes the gap as its sweeps grow (it needs ~1e6 sweeps for
1e-3: the reason the reference's default solver, not north_star's PGS, is what the engine implements -- DESIGN.md D1). Nothing reference-held pins contact dynamics
(DESIGN.md section 4), this is the strongest independent check available.

States: for `--scenes` IT1 scenes (4 boxes) and 6-object scenes (3 boxes + 3 spheres), the moments of an aimed grasp with the most contacts: fingers
closed on the object on the plate, the object lifted 40 steps off the plate, and mid-carry. Prints one JSON line, writes profiles/r03_solver_crosscheck.json.

    python tools/solver_crosscheck.py [--scenes 12]
"""
import argparse
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def grasp_states(model_spec, seed, rot):
    """States along one aimed grasp of scene `seed`: (tag, qpos, qvel, warmstart, pid, ctrl)."""
    from mujoco_rl_ur5_amd.model import load_model
    from oracle.oracle import Oracle
    m = load_model(model_spec)
    nobj = (m.nq - 8) // 7
    o = Oracle(m)
    o.reset(seed, 1, True)
    q = o.get_state()["qpos"]
    k = seed % nobj
    x, y = q[8 + 7 * k], -0.6 + q[8 + 7 * k + 1]
    o.move_ee([x, y, 1.1], 0.05, 1000)
    o.move_group(1 << 5, np.array([np.deg2rad([0, 30, 60, 90, -30, -60][rot])]), 0.05, 500)
    o.open_gripper(True)
    o.move_ee([x, y, 0.91], 0.01, 300)
    o.stay(100)
    out = []

    def grab(tag):
        s = o.get_state()
        out.append((tag, s["qpos"], s["qvel"], s["warmstart"], s["pid"], o.get_ctrl()))
    o.close_gripper(300)
    grab("closed")
    o.move_ee([0, -0.6, 1.1], 0.05, 40)
    grab("lifting")
    o.move_ee([0, -0.6, 1.1], 0.05, 1000)
    o.move_ee([0.6, 0.0, 1.15], 0.01, 120)
    grab("carrying")
    return m, out


def check(job):
    spec, seed, rot = job
    from oracle.oracle import Oracle
    m, states = grasp_states(spec, seed, rot)
    res = []
    for tag, qpos, qvel, warm, pid, ctrl in states:
        def solve(solver, iters, tol):
            o = Oracle(m)
            o.set_options(1, 0.0, solver)
            o.set_solver_limits(iters, tol)
            o.set_state(qpos=qpos, qvel=qvel, warmstart=warm, pid=pid)
            o.set_ctrl(ctrl)
            o.forward()
            x = o.vec("qacc")
            c, g = o.primal_cost(x)
            return x, c, g, o.solver_iter_last, len(o.contacts()), len(o.rows())
        xn, cn, gn, itn, ncon, nrow = solve(0, 0, -1.0)            # Newton as shipped: the model's iteration cap and tolerance
        sc = max(1.0, np.abs(xn).max())
        rec = dict(model=spec, seed=seed, state=tag, contacts=ncon, rows=nrow, newton_iterations=itn, qacc_max=float(np.abs(xn).max()),
                   newton_cost=cn, newton_gradient_norm=gn, pgs={})
        for sweeps in (100, 10000, 1000000):
            xp, cp, gp, itp, _, _ = solve(1, sweeps, 0.0)
            rec["pgs"][str(sweeps)] = dict(cost_above_newton=cp - cn, rel_distance_to_newton=float(np.abs(xp - xn).max() / sc), gradient_norm=gp)
        res.append(rec)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=12)
    ap.add_argument("--workers", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_solver_crosscheck.json"))
    a = ap.parse_args()
    jobs = [(spec, 20 + e, e % 2 * 3) for spec in ("it1_4box", "/UR5+gripper/UR5gripper_2_finger.xml") for e in range(a.scenes)]
    with ProcessPoolExecutor(a.workers) as ex:
        rows = [r for rs in ex.map(check, jobs) for r in rs]
    multi = [r for r in rows if r["contacts"] >= 4]
    gap = {k: [r["pgs"][k]["cost_above_newton"] for r in multi] for k in ("100", "10000", "1000000")}
    dist = {k: [r["pgs"][k]["rel_distance_to_newton"] for r in multi] for k in ("100", "10000", "1000000")}
    rep = dict(states=len(rows), multi_contact_states=len(multi), contacts_median=float(np.median([r["contacts"] for r in rows])), contacts_max=max(r["contacts"] for r in rows),
               rows_max=max(r["rows"] for r in rows), newton_iterations_max=max(r["newton_iterations"] for r in multi),
               newton_gradient_norm_max=max(r["newton_gradient_norm"] for r in multi),
               newton_gradient_norm_rel_max=max(r["newton_gradient_norm"] / max(1.0, abs(r["newton_cost"])) for r in multi),
               pgs_cost_never_below_newton=bool(min(min(v) for v in gap.values()) > -1e-7 * max(1.0, max(abs(r["newton_cost"]) for r in multi))),
               pgs_cost_gap_after_1e6_sweeps_below_gap_after_100=bool(all(c_ <= a_ for a_, c_ in zip(gap["100"], gap["1000000"]))),
               states_where_the_gap_is_not_monotone_in_between=int(sum(not (a_ >= b_ >= c_) for a_, b_, c_ in zip(gap["100"], gap["10000"], gap["1000000"]))),   # PGS descends on the dual; the primal cost read off its iterate need not be monotone
               pgs_cost_gap_median={k: float(np.median(v)) for k, v in gap.items()}, pgs_cost_gap_max={k: float(np.max(v)) for k, v in gap.items()},
               pgs_rel_distance_to_newton_median={k: float(np.median(v)) for k, v in dist.items()},
               pgs_rel_distance_to_newton_max={k: float(np.max(v)) for k, v in dist.items()})
    print(json.dumps(rep))
    rep["per_state"] = rows
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
