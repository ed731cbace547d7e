#!/usr/bin/env python3
"""What truncating the gripper's collision hulls changes, measured on the CPU oracle (which is generic in the hull size).

The reference collides the FULL convex hulls of robotiq_85_base_link_coarse / inner_knuckle_coarse / inner_finger_coarse
(UR5gripper_2_finger.xml:54-71,188-212: 400 / 70 / 120 hull vertices). Rounds 1-2 shipped 32-vertex approximations; the
round-2 verdict measured 11 of 96 reward bits changing. This tool compiles both small scenes with several caps, runs the same
aimed grasp attempts (check_mode 1) on each and reports the differences against the uncapped hulls.

    python tools/hull_cap_effect.py [--attempts 240] [--caps 32,64,128,0] [--out profiles/r03_hull_cap_effect.json]

Needs /root/reference (it compiles the MJCF); runs here, not on the GPU box.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = "/root/reference/UR5+gripper"


def _model(scene, cap):
    from mujoco_rl_ur5_amd.mjcf import compile_mjcf
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    if scene == "it1_4box":
        import importlib.util
        spec = importlib.util.spec_from_file_location("cm", os.path.join(ROOT, "tools", "compile_models.py"))
        src = open(spec.origin).read().split("jobs = [")[0]          # only the IT1_OBJECTS definition, not the compile loop
        ns = {"__file__": spec.origin}
        exec(compile(src, spec.origin, "exec"), ns)
        return compile_mjcf(os.path.join(REF, "UR5gripper_2_finger.xml"), objects=ns["IT1_OBJECTS"], maxhullvert=cap)
    return compile_mjcf(os.path.join(REF, "UR5gripper_2_finger.xml"), maxhullvert=cap)


def _run(job):
    scene, cap, e0, e1 = job
    from oracle.oracle import Oracle
    m = _model(scene, cap)
    nobj = (m.nq - 8) // 7
    out = []
    for e in range(e0, e1):
        o = Oracle(m)
        o.reset(20 + e, 1, True)
        objs = o.get_state()["qpos"][8:].reshape(-1, 7)
        k = e % nobj
        a = [objs[k, 0], -0.6 + objs[k, 1], 0.91]             # slide joints: world = body pos (0, -0.6, .) + offsets (tests/conftest.py aimed_actions)
        r, ps, pr = o.grasp_attempt(a, (e // nobj) % 6, 1)
        out.append((int(r), ps.tolist(), pr.tolist(), o.get_state()["qpos"].tolist()))
    return scene, cap, e0, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--attempts", type=int, default=240)
    ap.add_argument("--caps", default="32,64,128,0")
    ap.add_argument("--scenes", default="it1_4box,ur5_2f")
    ap.add_argument("--workers", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_hull_cap_effect.json"))
    a = ap.parse_args()
    caps = [int(c) for c in a.caps.split(",")]
    if 0 not in caps:
        caps.append(0)
    chunk = max(1, a.attempts // max(1, a.workers // 2))
    jobs = [(s, c, e, min(e + chunk, a.attempts)) for s in a.scenes.split(",") for c in caps for e in range(0, a.attempts, chunk)]
    t0 = time.time()
    res = {}
    with ProcessPoolExecutor(a.workers) as ex:
        for scene, cap, e0, out in ex.map(_run, jobs):
            res.setdefault((scene, cap), {})[e0] = out
    rep = {"attempts": a.attempts, "recipe": "Oracle.reset(20+e,1,True); aim at object e % nobj at z 0.91; grasp_attempt(a, (e//nobj) % 6, check_mode 1)", "scenes": {}}
    for scene in a.scenes.split(","):
        full = [x for e0 in sorted(res[(scene, 0)]) for x in res[(scene, 0)][e0]]
        rows = {}
        for cap in caps:
            got = [x for e0 in sorted(res[(scene, cap)]) for x in res[(scene, cap)][e0]]
            rew = sum(g[0] != f[0] for g, f in zip(got, full))
            steps = sum(g[1] != f[1] for g, f in zip(got, full))
            codes = sum(g[2] != f[2] for g, f in zip(got, full))
            dq = max(float(np.abs(np.array(g[3]) - np.array(f[3])).max()) for g, f in zip(got, full))
            rows[str(cap)] = dict(reward_bits_differ=rew, phase_step_counts_differ=steps, phase_results_differ=codes, max_abs_qpos_diff=dq,
                                  success_rate=float(np.mean([g[0] for g in got])))
        rep["scenes"][scene] = rows
    rep["wall_s"] = round(time.time() - t0, 1)
    print(json.dumps(rep, indent=1))
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
