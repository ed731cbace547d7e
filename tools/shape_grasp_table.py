#!/usr/bin/env python3
"""Per-shape grasp table: ONE object of UR5gripper_2_finger_many_objects.xml's pile (objects.xml:2-171: spheres, boxes, cylinders, capsules; condim 6,
margin 1e-3, solimp .99 .99 .01 from many_objects.xml:29) lying alone on the pick bin's floor, one full move_and_grasp script (GraspingEnv.py:205-386)
aimed at it for each wrist rotation. The reference's trained agent visibly holds such objects (media/gif_3.gif); this table shows which shapes / poses /
rotations the physics here can hold, on the CPU oracle and -- with --gpu on the GPU box -- on the HIP many-object engine for the same cases.

The grasp height is what GraspEnv.step derives from the depth image (GraspingEnv.py:100-104,258-259): the object's top surface at the aimed pixel,
then max(TABLE_HEIGHT = 0.91, z - 0.01). The bin floor of the pile scene is at 0.89 (many_objects.xml:120), so the fingertips stop 2 cm above it.

    python tools/shape_grasp_table.py [--gpu] [--out profiles/r03_shape_grasp_table.json]

Compiles MJCF when /root/reference exists; otherwise (GPU box) it needs the cached blobs of an earlier run under tools/probes/shape_models/.
"""
import argparse
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = "/root/reference/UR5+gripper/UR5gripper_2_finger_many_objects.xml"
CACHE = os.path.join(ROOT, "tools", "probes", "shape_models")
FLOOR = 0.89

S = float(np.sqrt(0.5))
# (label, type, size, poses): pose = (name, quaternion, height of the centre above the floor, height of the top surface above the centre)
SHAPES = []
for r in (0.02, 0.025, 0.03):
    SHAPES.append((f"sphere r{r}", "sphere", [r], [("resting", [1, 0, 0, 0], r, r)]))
for sz in ([0.015, 0.015, 0.015], [0.02, 0.02, 0.02], [0.025, 0.025, 0.025], [0.015, 0.02, 0.025], [0.025, 0.015, 0.025]):
    SHAPES.append((f"box {sz}", "box", sz, [("flat", [1, 0, 0, 0], sz[2], sz[2])]))
for r, h in ((0.015, 0.05), (0.02, 0.05), (0.025, 0.05), (0.025, 0.035)):
    for kind in ("cylinder", "capsule"):
        poses = [("lying along x", [S, 0, S, 0], r, r), ("lying along y", [S, S, 0, 0], r, r)]
        if kind == "cylinder":
            poses.append(("upright", [1, 0, 0, 0], h, h))
        SHAPES.append((f"{kind} r{r} h{h}", kind, [r, h], poses))


def model_for(idx):
    from mujoco_rl_ur5_amd.model import CompiledModel
    path = os.path.join(CACHE, f"shape_{idx}.ur5m")
    if os.path.exists(REF):
        from mujoco_rl_ur5_amd.mjcf import compile_mjcf
        label, typ, size, _ = SHAPES[idx]
        m = compile_mjcf(REF, objects=[dict(name="object_0", type=typ, size=size, pos=[0.0, -0.6, 0.95], joints="free")])
        os.makedirs(CACHE, exist_ok=True)
        m.save(path)
        return m
    return CompiledModel.load(path)


def case_state(o_or_none, m, pose):
    _, quat, zc, ztop = pose
    return np.array([0.0, -0.6, FLOOR + zc - 5e-5, *quat]), FLOOR + zc + ztop


def run_oracle(job):
    idx, pi, rot = job
    from oracle.oracle import Oracle
    m = model_for(idx)
    o = Oracle(m)
    o.reset(20, 1, False)
    st = o.get_state()
    q7, _ = case_state(o, m, SHAPES[idx][3][pi])
    st["qpos"][8:15] = q7
    o.set_state(qpos=st["qpos"], qvel=np.zeros(m.nv))
    o.stay(1000)
    q = o.get_state()["qpos"]
    top = q[10] + SHAPES[idx][3][pi][3]
    r, ps, pr = o.grasp_attempt([q[8], q[9], top], rot, 0)
    qf = o.get_state()["qpos"]
    return idx, pi, rot, int(r), ps.tolist(), pr.tolist(), q[8:15].tolist(), qf[8:11].tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true", help="also run every case on the HIP many-object engine (one scene per case)")
    ap.add_argument("--workers", type=int, default=min(64, os.cpu_count() or 1))
    ap.add_argument("--rots", default="0,1,2,3,4,5")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_shape_grasp_table.json"))
    a = ap.parse_args()
    rots = [int(x) for x in a.rots.split(",")]
    for i in range(len(SHAPES)):
        model_for(i)                                            # compile / cache before the workers start
    jobs = [(i, pi, rot) for i, sh in enumerate(SHAPES) for pi in range(len(sh[3])) for rot in rots]
    with ProcessPoolExecutor(a.workers) as ex:
        res = list(ex.map(run_oracle, jobs, chunksize=2))
    table = {}
    for idx, pi, rot, r, ps, pr, q0, qf in res:
        key = f"{SHAPES[idx][0]} | {SHAPES[idx][3][pi][0]}"
        table.setdefault(key, {"oracle": {}, "oracle_close_steps": {}})
        table[key]["oracle"][str(rot)] = r
        table[key]["oracle_close_steps"][str(rot)] = [ps[5], ps[9]]       # grasp() at the object (max 300), closing check at the drop position (max 1000)
    if a.gpu:
        from mujoco_rl_ur5_amd.native import BatchSim
        for i, sh in enumerate(SHAPES):
            m = model_for(i)
            cases = [(pi, rot) for pi in range(len(sh[3])) for rot in rots]
            sim = BatchSim(m, len(cases))
            sim.reset(np.full(len(cases), 20, dtype=np.uint64), 1, 0.0)
            st = sim.get_state()
            for c, (pi, rot) in enumerate(cases):
                st["qpos"][c, 8:15], _ = case_state(None, m, sh[3][pi])
            sim.set_state(qpos=st["qpos"], qvel=np.zeros_like(st["qvel"]))
            sim.stay(1000.0)
            q = sim.get_state()["qpos"]
            acts = np.array([[q[c, 8], q[c, 9], q[c, 10] + sh[3][pi][3]] for c, (pi, rot) in enumerate(cases)])
            rew, ps, pr = sim.grasp_attempt(acts, rot=np.array([rot for _, rot in cases]), check_mode=0)
            if sim.variant != 1 or sim.counters()["status"].max() != 0:
                print(f"# {sh[0]}: engine variant {sim.variant}, status bits {sim.counters()['status'].tolist()}", file=sys.stderr)
            for c, (pi, rot) in enumerate(cases):
                key = f"{sh[0]} | {sh[3][pi][0]}"
                table[key].setdefault("gpu", {})[str(rot)] = int(rew[c])
                table[key].setdefault("gpu_close_steps", {})[str(rot)] = [int(ps[c][5]), int(ps[c][9])]
    n = len(res)
    rep = {"cases": n, "rotations_deg": {"0": 0, "1": 30, "2": 60, "3": 90, "4": -30, "5": -60}, "floor_z": FLOOR, "table_height": 0.91,
           "oracle_success_rate": float(np.mean([r[3] for r in res])), "table": table}
    if a.gpu:
        agree = sum(int(v["gpu"][k] == v["oracle"][k]) for v in table.values() for k in v["oracle"])
        pos = [(v["oracle"][k], v["gpu"][k]) for v in table.values() for k in v["oracle"]]
        rep["gpu_success_rate"] = float(np.mean([g for _, g in pos]))
        rep["gpu_agreement"] = agree / n
        rep["agreement_on_oracle_positives"] = float(np.mean([g for o_, g in pos if o_ == 1])) if any(o_ for o_, _ in pos) else None
    print(json.dumps({k: v for k, v in rep.items() if k != "table"}))
    for k, v in table.items():
        print(f"{k:42s} oracle {[v['oracle'][str(r)] for r in rots]}" + (f"  gpu {[v['gpu'][str(r)] for r in rots]}" if a.gpu else "")
              + f"  close/check steps {[v['oracle_close_steps'][str(r)] for r in rots]}")
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
