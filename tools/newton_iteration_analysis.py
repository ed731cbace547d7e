#!/usr/bin/env python3
"""Why does a SETTLED 40-object pile need ~11 Newton iterations per step?  (round-3 verdict, item 2 ii; CPU only: oracle + the numpy rows of oracle/refrows.py)

For every step of a settled pile (reset + 1000 ms settle, then `steps` more steps with the arm at rest): the solver's iteration count, the contacts that appeared /
disappeared since the previous step, and the constraint rows whose state (active <-> inactive) differs between the warm start (the previous step's acceleration,
MuJoCo's `qacc_warmstart` [3P]) and the solution. Newton on this piecewise-quadratic problem is exact for a FIXED active set: an iteration that changes no row's
state ends the solve, so the iteration count is ~ (rounds of active-set changes) + 2. Prints one JSON line.
    python tools/newton_iteration_analysis.py [seed=31] [steps=60] > profiles/r04_newton_iterations_pile.json"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from oracle import refrows
from oracle.oracle import Oracle

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 31
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
o = Oracle(m)
o.reset(seed, 1, True)
prev_pairs = None
rec = []
for k in range(steps):
    st = o.get_state()
    o.forward()
    con = o.contacts()
    R = refrows.build_rows(m, st["qpos"], st["qvel"], con)
    qacc, warm = o.vec("qacc"), st["warmstart"]
    jar_w, jar_s = R.J @ warm - R.aref, R.J @ qacc - R.aref
    act_w, act_s = (~R.unilateral) | (jar_w < 0), (~R.unilateral) | (jar_s < 0)
    pairs = {(int(c[7]), int(c[8])) for c in con}
    depth = np.array([c[0] for c in con]) - 1e-3                                # dist - margin: < 0 inside the 1 mm contact margin
    rec.append(dict(iterations=o.solver_iter_last, contacts=len(con), rows=len(R.pos), active_at_solution=int(act_s.sum()),
                    rows_whose_state_differs_from_the_warm_start=int((act_w != act_s).sum()),
                    pairs_appeared=0 if prev_pairs is None else len(pairs - prev_pairs), pairs_disappeared=0 if prev_pairs is None else len(prev_pairs - pairs),
                    contacts_within_1e_5_of_the_margin=int((np.abs(depth) < 1e-5).sum()), max_qvel=float(np.abs(st["qvel"]).max())))
    prev_pairs = pairs
    o.step(1)
it = np.array([r["iterations"] for r in rec])
fl = np.array([r["rows_whose_state_differs_from_the_warm_start"] for r in rec])
print(json.dumps(dict(scene="40-object pile, seed %d, settled 1000 ms, arm at rest" % seed, steps=steps,
                      iterations_mean=float(it.mean()), iterations_histogram={str(v): int((it == v).sum()) for v in sorted(set(it.tolist()))},
                      contacts_mean=float(np.mean([r["contacts"] for r in rec])), rows_mean=float(np.mean([r["rows"] for r in rec])),
                      active_rows_mean=float(np.mean([r["active_at_solution"] for r in rec])),
                      rows_differing_from_warm_start_mean=float(fl.mean()), rows_differing_from_warm_start_max=int(fl.max()),
                      correlation_iterations_vs_differing_rows=float(np.corrcoef(it, fl)[0, 1]) if it.std() > 0 and fl.std() > 0 else None,
                      contact_pairs_appearing_per_step=float(np.mean([r["pairs_appeared"] for r in rec[1:]])),
                      contact_pairs_disappearing_per_step=float(np.mean([r["pairs_disappeared"] for r in rec[1:]])),
                      contacts_within_1e_5_m_of_the_margin_mean=float(np.mean([r["contacts_within_1e_5_of_the_margin"] for r in rec])),
                      max_abs_qvel_mean=float(np.mean([r["max_qvel"] for r in rec])), per_step=rec)))
