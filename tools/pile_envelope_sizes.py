#!/usr/bin/env python3
"""How large does the envelope of the pile's Newton Hessian get?  (round 5; CPU only: oracle states + the numpy restatement of envelope_structure() in
tools/pile_structure_stats.py.) The LDS pool that holds the envelope since round 5 has 2 384 doubles (csrc/ur5_engine.h Lds::HENV_DOUBLES); this samples the drop / settle of
n piles and a scripted approach - descend - close - lift on each, every 10-20 steps, and prints the size of the whole envelope and of its coupled blocks alone per phase.
    python tools/pile_envelope_sizes.py [n_piles=4] > profiles/r05_pile_envelope_sizes.log"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from mujoco_rl_ur5_amd.model import load_model
from oracle.oracle import Oracle
from pile_structure_stats import blocks_of_bodies, structure
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
blk, nobj, obj_body = blocks_of_bodies(m)
def stat(o):
    o.forward()
    cons = o.contacts()
    pairs = []
    for c in cons:
        a, b = blk[int(m.geom_bodyid[int(c[7])])], blk[int(m.geom_bodyid[int(c[8])])]
        if a >= 0 and b >= 0 and a != b: pairs.append((a, b))
    x = o.body_xpos()[obj_body, 0]
    s = structure(nobj, x, pairs)
    # coupled-only size: env minus 21 (or 36) per single block
    singles = (nobj + 1) - s["coupled"]
    robot_single = 1 if not any(nobj in p for p in pairs) else 0
    return len(cons), s["env"], s["env"] - 21 * (singles - robot_single) - 36 * robot_single, s["coupled"], max(s["nr"].values()) if s["nr"] else 0
rows = []
for e in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    o = Oracle(m); o.reset(20 + e, 1, False)
    for k in range(50):           # drop + settle, sampled every 20 steps
        o.step(10); rows.append(("settle",) + stat(o))
    st = o.get_state()
    objs = st["qpos"][8:].reshape(-1, 7)
    inb = [k for k in range(nobj) if abs(objs[k,0]) < 0.15 and abs(objs[k,1] + 0.6) < 0.1]
    k = max(inb, key=lambda k: objs[k,2]) if inb else 0
    tgt = np.array([objs[k,0], objs[k,1], objs[k,2] - 0.01])
    o.open_gripper(False) if hasattr(o, "open_gripper") else None
    for phase, xyz in (("above", tgt + [0,0,0.25]), ("descend", tgt), ):
        for _ in range(25):
            r, n = o.move_ee(xyz, 0.01, 20); rows.append((phase,) + stat(o))
            if r == 0: break
    for _ in range(15):
        o.close_gripper(20); rows.append(("close",) + stat(o))
    for _ in range(25):
        r, n = o.move_ee(tgt + [0,0,0.3], 0.01, 20); rows.append(("lift",) + stat(o))
        if r == 0: break
import collections
by = collections.defaultdict(list)
for r in rows: by[r[0]].append(r[1:])
for ph, v in by.items():
    v = np.array(v)
    print("%-8s n=%3d  contacts mean %.0f max %d | envelope (all blocks) mean %.0f p90 %.0f max %d | coupled-only mean %.0f p90 %.0f max %d | coupled blocks mean %.1f max %d | rows reaching a panel max %d" % (
        ph, len(v), v[:,0].mean(), v[:,0].max(), v[:,1].mean(), np.percentile(v[:,1], 90), v[:,1].max(), v[:,2].mean(), np.percentile(v[:,2], 90), v[:,2].max(), v[:,3].mean(), v[:,3].max(), v[:,4].max()))
allv = np.array([r[1:] for r in rows])
print("fraction of sampled steps with the whole envelope <= 2384: %.3f; coupled-only <= 2384: %.3f, <= 1808: %.3f, <= 1152: %.3f" % ((allv[:,1] <= 2384).mean(), (allv[:,2] <= 2384).mean(), (allv[:,2] <= 1808).mean(), (allv[:,2] <= 1152).mean()))
