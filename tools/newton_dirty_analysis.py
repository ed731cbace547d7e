#!/usr/bin/env python3
"""How much of the pile's Newton Hessian changes from one iteration to the next?  (round-4 verdict, item 1b; CPU only: oracle + its Newton trace hook)

The pile kernel re-assembles and refactors the whole envelope whenever ANY constraint row changed its state since the previous iteration. This tool records, for every
Newton iteration of `steps` steps of settled piles, which contacts changed their active mask, maps them to Hessian blocks through the numpy restatement of
envelope_structure() (tools/pile_structure_stats.py) and asks what a PARTIAL refactorisation would still have to touch: an envelope group (a maximal block range no
row crosses: independent chains of the factorisation) is dirty when one of its blocks has a changed contact, and inside a dirty group the factor columns left of the
first dirty block are unchanged. Prints one JSON line.
    python tools/newton_dirty_analysis.py [n_piles=4] [steps=30] [settle_ms=1000]"""
import json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from mujoco_rl_ur5_amd.model import load_model
from oracle.oracle import Oracle
from pile_structure_stats import blocks_of_bodies

n_piles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
settle = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
blk, nobj, obj_body = blocks_of_bodies(m)
nblk = nobj + 1


def structure(x, pairs):
    """sorted position of every block, first coupled block, envelope groups (as envelope_structure())."""
    island = np.arange(nblk)
    for _ in range(6):
        new = island.copy()
        for a, b in pairs:
            la, lb = island[a], island[b]
            if la < lb: new[a] = max(new[a], lb)
            elif lb < la: new[b] = max(new[b], la)
        island = new[new]
    order = sorted(range(nobj), key=lambda k: (island[k], x[k], k))
    rank = np.zeros(nblk, dtype=int); rank[order] = np.arange(nobj); rank[nobj] = nobj
    first = np.arange(nblk)
    for a, b in pairs:
        pa, pb = sorted((rank[a], rank[b]))
        first[pb] = min(first[pb], pa)
    last = np.arange(nblk)
    for p in range(nblk):
        for q in range(p + 1, nblk):
            if first[q] <= p: last[p] = q
    grp = np.zeros(nblk, dtype=int); g, end = -1, -1
    for p in range(nblk):
        if p > end: g += 1
        end = max(end, last[p]); grp[p] = g
    return rank, first, last, grp


def cost_of(first, last, lo_by_group, grp):
    """cost proxy of a (partial) factorisation: sum over the panels that are refactored of (rows reaching the panel)^2 (trailing update) + passes: here simply the
    number of panels with rows below them and the number of envelope entries right of the restart column."""
    panels, entries = 0, 0
    for p in range(nblk):
        lo = lo_by_group.get(grp[p])
        if lo is None or p < lo: continue
        if last[p] != p: panels += 1
        w = 6 if p < nobj else 8
        entries += w * (6 * (p - max(first[p], lo))) + w * (w + 1) // 2
    return panels, entries


by_it = {}
tot = dict(evals=0, refactor=0)
for e in range(n_piles):
    o = Oracle(m)
    o.reset(20 + e, 1, False)
    o.stay(settle)
    o.newton_trace(True)
    for k in range(steps):
        o.forward()
        con = o.contacts()
        act, rc = o.get_newton_trace()
        o.step(1)
        if len(act) == 0: continue
        cb = []
        for c in con:
            a, b = blk[int(m.geom_bodyid[int(c[7])])], blk[int(m.geom_bodyid[int(c[8])])]
            cb.append((a, b))
        pairs = [(a, b) for a, b in cb if a >= 0 and b >= 0 and a != b]
        x = o.body_xpos()[obj_body, 0]
        rank, first, last, grp = structure(x, pairs)
        ngroups_coupled = len({grp[p] for p in range(nblk) if first[p] != p or last[p] != p})
        full = cost_of(first, last, {g: 0 for g in set(grp)}, grp)
        for it in range(1, len(act)):
            ch_rows = np.flatnonzero(act[it] != act[it - 1])
            tot["evals"] += 1
            d = by_it.setdefault(min(it, 12), dict(n=0, refactor=0, rows=0, contacts=0, blocks=0, groups=0, coupled_groups=0, panels=0, entries=0, full_panels=0, full_entries=0, touch_coupled=0))
            d["n"] += 1
            if len(ch_rows) == 0: continue
            tot["refactor"] += 1; d["refactor"] += 1
            ch_con = sorted({int(rc[r]) for r in ch_rows if rc[r] >= 0})
            dirty_blocks = set()
            for r in ch_rows:
                if rc[r] < 0: dirty_blocks.add(nobj)                       # equality / limit rows live in the robot block
            for c in ch_con:
                a, b = cb[c]
                for bb in (a, b):
                    if bb >= 0: dirty_blocks.add(int(rank[bb]))
            lo = {}
            for p in dirty_blocks: lo[grp[p]] = min(lo.get(grp[p], 10 ** 9), p)
            # a dirty block in the middle of a group: columns left of it keep their factor; everything right of it in the group is refactored
            pn, en = cost_of(first, last, lo, grp)
            d["rows"] += len(ch_rows); d["contacts"] += len(ch_con); d["blocks"] += len(dirty_blocks); d["groups"] += len(lo)
            d["coupled_groups"] += ngroups_coupled; d["panels"] += pn; d["entries"] += en; d["full_panels"] += full[0]; d["full_entries"] += full[1]
            d["touch_coupled"] += int(any(first[p] != p or last[p] != p for p in dirty_blocks))
out = dict(scene="%d 40-object piles settled %.0f ms, %d steps each, arm at rest" % (n_piles, settle, steps), hessian_evaluations_after_the_first=tot["evals"],
           evaluations_with_a_changed_row=tot["refactor"], by_iteration={})
for it in sorted(by_it):
    d = by_it[it]; r = max(1, d["refactor"])
    out["by_iteration"][str(it) + ("+" if it == 12 else "")] = dict(evaluations=d["n"], with_changed_rows=d["refactor"], rows_changed=d["rows"] / r, contacts_changed=d["contacts"] / r,
        blocks_dirty=d["blocks"] / r, groups_dirty=d["groups"] / r, coupled_groups=d["coupled_groups"] / r, touches_a_coupled_block=d["touch_coupled"] / r,
        panels_to_refactor=d["panels"] / r, panels_full=d["full_panels"] / r, entries_to_refactor=d["entries"] / r, entries_full=d["full_entries"] / r)
print(json.dumps(out))
