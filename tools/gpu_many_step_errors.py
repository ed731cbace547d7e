#!/usr/bin/env python3
"""One physics step of settled piles on the HIP kernel and on the oracle FROM THE SAME STATE: how far apart are they after one step, and do they take the same number of
Newton iterations?  (round 5: the size of the first difference is what sets WHEN the two trajectories cross 1e-6, tools/pile_divergence_time.py.)
    python tools/gpu_many_step_errors.py <states.npz> [scenes=64] [steps=5]"""
import json, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from oracle.oracle import Oracle

D = np.load(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
sim = BatchSim(m, n)
st = dict(qpos=D["qpos"][:n].copy(), qvel=D["qvel"][:n].copy(), warmstart=D["warmstart"][:n].copy(), pid=D["pid"][:n].copy())
ctrl = D["ctrl"][:n].copy()
orc = [Oracle(m) for _ in range(n)]
rows = []
for k in range(steps):                                                         # step by step along the ORACLE's trajectory: both sides start every step from the same state
    sim.set_state(**st)
    sim.set_ctrl(ctrl)
    it0 = sim.counters()["solver_iters"].copy()
    sim.step(1)
    g = sim.get_state()
    git = sim.counters()["solver_iters"] - it0

    def one(e):
        o = orc[e]
        o.set_state(qpos=st["qpos"][e], qvel=st["qvel"][e], warmstart=st["warmstart"][e], pid=st["pid"][e])
        o.set_ctrl(ctrl[e])
        o.step(1)
        s = o.get_state()
        return s["qpos"], s["qvel"], s["warmstart"], o.solver_iter_last
    with ThreadPoolExecutor(max_workers=min(n, os.cpu_count() or 8)) as ex:
        res = list(ex.map(one, range(n)))
    oq, ov, ow, oit = (np.stack([r[i] for r in res]) for i in range(4))
    dq, dv = np.abs(g["qpos"] - oq).max(axis=1), np.abs(g["qvel"] - ov).max(axis=1)
    da = np.abs(g["warmstart"] - ow).max(axis=1) / np.maximum(1.0, np.abs(ow).max(axis=1))
    same = git == oit
    jumps = {"above_1e-14": int((dq > 1e-14).sum()), "above_1e-12": int((dq > 1e-12).sum()), "above_1e-10": int((dq > 1e-10).sum()), "above_1e-8": int((dq > 1e-8).sum())}
    rows.append(dict(step=k, scenes_by_qpos_diff=jumps, qpos_diff_median=float(np.median(dq)), qpos_diff_max=float(dq.max()), qvel_diff_median=float(np.median(dv)), qacc_rel_diff_median=float(np.median(da)),
                     qacc_rel_diff_max=float(da.max()), scenes_with_equal_newton_iterations=int(same.sum()), qpos_diff_median_equal_iterations=float(np.median(dq[same])) if same.any() else None,
                     qpos_diff_median_other_iterations=float(np.median(dq[~same])) if (~same).any() else None, kernel_iterations_mean=float(git.mean()), oracle_iterations_mean=float(oit.mean())))
    st = dict(qpos=oq, qvel=ov, warmstart=ow, pid=st["pid"])
print(json.dumps(dict(scenes=n, steps=steps, per_step=rows)))
