#!/usr/bin/env python3
"""The arm-link hulls on the reference's DEFAULT scene (40-object piles, UR5gripper_2_finger_many_objects.xml): do they change anything?

Same question as tools/arm_collision_effect.py, for the scene in which objects are dropped from z = 1.0 .. 1.5 m over the bin while the arm
hovers above it (GraspingEnv.py:418-430): oracle with and without the seven arm hulls (capped at 32 vertices), per scene the reset + settle
and one random-agent grasp attempt. Reports how many settled piles / attempts differ and whether any arm hull ever carried a contact.
Needs /root/reference; writes profiles/<tag>_arm_collision_effect_piles.json.   python tools/arm_collision_effect_piles.py [n] [tag]
"""
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = "/root/reference/UR5+gripper/UR5gripper_2_finger_many_objects.xml"


def models():
    import mujoco_rl_ur5_amd.mjcf as mj
    m_off = mj.compile_mjcf(SRC)
    m_on = mj.compile_mjcf(SRC, arm_collision=True)
    return m_off, m_on


def one(e):
    from mujoco_rl_ur5_amd.controller import MJ_Controller
    from oracle.oracle import Oracle
    m_off, m_on = models()

    class _NoSim:
        n = 1
    cam = MJ_Controller(m_off, simulation=_NoSim())
    rng = np.random.default_rng(1000 + e)
    px, py, rot = int(rng.integers(0, 200)), int(rng.integers(0, 200)), int(rng.integers(0, 6))
    w = cam.pixel_2_world(px, py, 2.0 - 0.89)
    out = []
    for m in (m_off, m_on):
        o = Oracle(m)
        o.reset(20 + e, 1, True)
        q_settled = o.qpos.copy()
        r = o.grasp_attempt([w[0], w[1], 0.89], rot, 0, 0.89)
        out.append((q_settled, r, o.qpos.copy()))
    (sa, ra, qa), (sb, rb, qb) = out
    return dict(scene=e, settle_diff=float(np.abs(sa - sb).max()), reward=[int(ra[0]), int(rb[0])],
                steps_equal=bool(np.array_equal(ra[1], rb[1])), results_equal=bool(np.array_equal(ra[2], rb[2])),
                final_diff=float(np.abs(qa - qb).max()))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
    t0 = time.time()
    with Pool(min(n, os.cpu_count())) as p:
        rows = p.map(one, range(n))
    res = dict(n_scenes=n, settled_piles_that_differ=sum(r["settle_diff"] > 0 for r in rows),
               attempts_that_differ_in_reward=sum(r["reward"][0] != r["reward"][1] for r in rows),
               attempts_that_differ_in_steps=sum(not r["steps_equal"] for r in rows),
               attempts_that_differ_in_results=sum(not r["results_equal"] for r in rows), rows=rows, seconds=round(time.time() - t0, 1))
    with open(os.path.join(ROOT, "profiles", f"{tag}_arm_collision_effect_piles.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "rows"}))
