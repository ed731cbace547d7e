#!/usr/bin/env python3
"""What the pile kernel's envelope factorisation has to work with, on the CPU: the coupling structure of settled 40-object piles (oracle states), pushed through an
independent numpy restatement of csrc/ur5_engine.h envelope_structure() -- islands, block order (island, then x, robot last), first coupled block, envelope groups,
levels, rows reaching every panel. Prints per pile: coupled contacts, coupled blocks, envelope size (checked against the engine's own figure through the lane
emulation's ur5_forward_debug), levels, panels per level, passes of four panels, rows / row pairs per panel.
    python tools/pile_structure_stats.py [n_piles] [settle_ms]"""
import os, sys
import ctypes as C
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim, _VARIANT
from oracle.oracle import Oracle


def blocks_of_bodies(m):
    """body id -> block: object k (free joint k, in joint order) -> k, a body moved by a hinge / slide joint (the robot) -> nobj, static -> -1"""
    free = [j for j in range(len(m.jnt_type)) if m.jnt_type[j] == 0]
    nobj = len(free)
    blk = -np.ones(m.nbody, dtype=int)
    for k, j in enumerate(free):
        blk[m.jnt_bodyid[j]] = k
    robot_joint_bodies = {int(m.jnt_bodyid[j]) for j in range(len(m.jnt_type)) if m.jnt_type[j] != 0}
    for b in range(1, m.nbody):
        a = b
        while a > 0:
            if a in robot_joint_bodies:
                blk[b] = nobj
                break
            if blk[a] >= 0 and blk[a] < nobj:
                break
            a = int(m.body_parentid[a])
    return blk, nobj, [int(m.jnt_bodyid[j]) for j in free]


def structure(nobj, x, pairs):
    """pairs: (block a, block b) of every contact between two movable bodies (objects 0..nobj-1, robot = nobj). Returns the lists of envelope_structure()."""
    nblk = nobj + 1
    island = np.arange(nblk)
    for _ in range(6):                                   # label propagation with pointer jumping, label = largest member
        new = island.copy()
        for a, b in pairs:
            la, lb = island[a], island[b]
            if la < lb: new[a] = max(new[a], lb)
            elif lb < la: new[b] = max(new[b], la)
        island = new[new]
    order = sorted(range(nobj), key=lambda k: (island[k], x[k], k))      # island, then x
    rank = np.zeros(nobj, dtype=int); rank[order] = np.arange(nobj)
    pos = lambda blk: nobj if blk == nobj else rank[blk]                 # sorted position; the robot block is last
    first = np.arange(nblk)
    for a, b in pairs:
        pa, pb = sorted((pos(a), pos(b)))
        first[pb] = min(first[pb], pa)
    last = np.arange(nblk); reach = [[] for _ in range(nblk)]
    for p in range(nblk):
        for q in range(p + 1, nblk):
            if first[q] <= p:
                last[p] = q; reach[p].append(q)
    width = lambda p: 6 if p < nobj else 8
    env = sum(sum(6 * (p - first[p]) + j + 1 for j in range(width(p))) for p in range(nblk))
    lv = -np.ones(nblk, dtype=int); grp_end, posn = -1, 0
    for p in range(nblk):
        if p > grp_end: posn = 0
        grp_end = max(grp_end, last[p])
        if last[p] != p:
            lv[p] = posn; posn += 1
    coupled = sum(1 for p in range(nblk) if first[p] != p or last[p] != p)
    nr = {p: sum(width(q) for q in reach[p]) for p in range(nblk) if lv[p] >= 0}
    touched = {a for a, b in pairs} | {b for a, b in pairs}
    return dict(env=env, coupled=coupled, lv=lv, nr=nr, islands=len({int(island[b]) for b in touched}))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    settle = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
    m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
    blk, nobj, obj_body = blocks_of_bodies(m)
    import conftest
    sim = BatchSim(m, 1, lib_path=conftest.build_emul())
    stride = _VARIANT[sim.variant][1]
    tot = dict(levels=[], panels=[], passes=[], nr=[], pairs=[], env=[], coupled=[])
    for e in range(n):
        o = Oracle(m)
        o.reset(20 + e, 1, False)
        o.stay(settle)
        o.forward()
        st = o.get_state()
        cons = o.contacts()
        pairs = []
        for c in cons:
            a, b = blk[int(m.geom_bodyid[int(c[7])])], blk[int(m.geom_bodyid[int(c[8])])]
            if a >= 0 and b >= 0 and a != b:
                pairs.append((a, b))
        x = o.body_xpos()[obj_body, 0]
        s = structure(nobj, x, pairs)
        sim.set_state(qpos=st["qpos"][None], qvel=st["qvel"][None], warmstart=st["warmstart"][None], pid=st["pid"][None])
        out = np.zeros((1, stride))
        sim.lib.ur5_forward_debug(sim._h, out.ctypes.data_as(C.POINTER(C.c_double)))
        eng_env, eng_ncouple, eng_coupled = int(out[0, 4]), int(out[0, 5]), int(out[0, 6])
        lv = s["lv"]; nl = lv.max() + 1
        per_level = [int((lv == l).sum()) for l in range(nl)]
        passes = sum((k + 3) // 4 for k in per_level)
        nrs = list(s["nr"].values())
        print("pile %2d: %2d contacts, %2d between movable bodies (engine %2d), %2d coupled blocks (engine %2d) in %d islands, envelope %5d doubles (engine %5d)%s; %2d levels, panels per level %s -> %2d passes; "
              "rows reaching a panel: mean %.1f max %d; row pairs per panel: mean %.0f max %d" % (
                  e, len(cons), len(pairs), eng_ncouple, s["coupled"], eng_coupled, s["islands"], s["env"], eng_env, "" if (s["env"], s["coupled"], len(pairs)) == (eng_env, eng_coupled, eng_ncouple) else "  MISMATCH",
                  nl, per_level, passes, np.mean(nrs) if nrs else 0, max(nrs) if nrs else 0, np.mean([r * (r + 1) / 2 for r in nrs]) if nrs else 0, max([r * (r + 1) // 2 for r in nrs]) if nrs else 0))
        tot["levels"].append(nl); tot["passes"].append(passes); tot["nr"] += nrs; tot["env"].append(s["env"]); tot["coupled"].append(s["coupled"])
    print("all %d piles after %.0f ms: levels %.1f (max %d), passes per factorisation %.1f, rows reaching a panel %.1f (90th percentile %.0f, max %d), envelope %.0f doubles, coupled blocks %.1f" % (
        n, settle, np.mean(tot["levels"]), max(tot["levels"]), np.mean(tot["passes"]), np.mean(tot["nr"]), np.percentile(tot["nr"], 90), max(tot["nr"]), np.mean(tot["env"]), np.mean(tot["coupled"])))


if __name__ == "__main__":
    main()
