#!/usr/bin/env python3
"""What do the seven arm-link collision hulls change?

The oracle is generic in the pair list, so it can run the scene compiled WITH the arm hulls (mjcf.compile_mjcf(arm_collision=True), hulls
capped at 32 vertices like the gripper's) next to the shipped one. For N scenes: reset + settle, then one grasp attempt at a pixel drawn
uniformly from the WHOLE 200x200 image (the reference's random agent, example_agent.py: action_space.sample()), z from the fixed table
height. Reports how many attempts differ in reward, in any phase result, or in any phase step count, and the worst qpos difference.
Needs /root/reference (compiles the MJCF); writes profiles/<tag>_arm_collision_effect.json.

    python tools/arm_collision_effect.py [n_scenes] [tag]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mujoco_rl_ur5_amd.mjcf as mj  # noqa: E402
from mujoco_rl_ur5_amd.controller import MJ_Controller  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = "/root/reference/UR5+gripper/UR5gripper_2_finger.xml"
objs = [dict(name=f"box_{k + 1}", type="box", size=[0.02, 0.02, 0.02], pos=[0.0, -0.6, 0.95 + 0.1 * k], joints="slide3ball",
             rgba=(0.5, 0.5, 0.5, 1)) for k in range(4)]
m_off = mj.compile_mjcf(src, objects=objs)
m_on = mj.compile_mjcf(src, objects=objs, arm_collision=True)


class _NoSim:
    n = 1


cam = MJ_Controller(m_off, simulation=_NoSim())
rng = np.random.default_rng(0)
diff_reward = diff_result = diff_steps = 0
worst = 0.0
arm_contacts = 0
rewards = [0, 0]
t0 = time.time()
cases = []
for e in range(n):
    px, py = int(rng.integers(0, 200)), int(rng.integers(0, 200))
    w = cam.pixel_2_world(px, py, 2.0 - 0.91)
    out = []
    for k, m in enumerate((m_off, m_on)):
        o = Oracle(m)
        o.reset(20 + e, 1, True)
        r = o.grasp_attempt([w[0], w[1], 0.91], 0, 0, 0.91)
        out.append((r, o.qpos.copy()))
        rewards[k] += int(r[0])
    (ra, qa), (rb, qb) = out
    dr, ds, dres = int(ra[0] != rb[0]), int(not np.array_equal(ra[1], rb[1])), int(not np.array_equal(ra[2], rb[2]))
    diff_reward += dr; diff_steps += ds; diff_result += dres
    worst = max(worst, float(np.abs(qa - qb).max()))
    if dr or ds or dres:
        cases.append(dict(scene=e, pixel=[px, py], world=[float(w[0]), float(w[1])], steps_off=ra[1].tolist(), steps_on=rb[1].tolist(),
                          results_off=ra[2].tolist(), results_on=rb[2].tolist(), reward=[int(ra[0]), int(rb[0])]))
res = dict(n_attempts=n, rule="pixel uniform over the whole 200x200 image, z = 0.91, rotation 0", npair_off=int(len(m_off.pair_geom1)),
           npair_on=int(len(m_on.pair_geom1)), differ_in_reward=diff_reward, differ_in_phase_result=diff_result,
           differ_in_phase_steps=diff_steps, max_abs_qpos_difference=worst, rewards_off_on=rewards, differing_cases=cases[:20],
           seconds=round(time.time() - t0, 1))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", f"{tag}_arm_collision_effect.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "differing_cases"}))
