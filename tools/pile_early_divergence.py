#!/usr/bin/env python3
"""The first `horizon` steps of the pile divergence statistic, CPU side in a minute instead of an hour: the grasp script starts with move_ee(above the target, 0.05, 1000)
(GraspingEnv.py:212), which lasts several hundred steps, so the oracle only has to run that phase capped at `horizon` steps to give the trajectory samples of every
checkpoint <= horizon. Prints, per pair, the distribution of max |dqpos| at each early checkpoint and the first checkpoint above 1e-6.
    python tools/pile_early_divergence.py <states.npz> <divergence_gpu.npz | oracle:VARIANT[:CHOLESKY]> [horizon=120] [scenes=256] [threads=4] [oracle variant: "" | fma | fmadyn]
With `oracle:fmadyn` (or `oracle::2`, `oracle:fma`, ...) in place of the kernel's file the first side is a control twin of the oracle instead of the HIP kernel."""
import json, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from oracle.oracle import Oracle

D = np.load(sys.argv[1])
ALLCK = [5, 10, 20, 30, 40, 60, 80, 100, 120, 160, 240, 320, 400, 480, 560, 640, 800, 1000, 1200, 1500, 1800]
TWIN = sys.argv[2].split(":") if sys.argv[2].startswith("oracle") else None
G = dict(checkpoints=np.array(ALLCK), qpos=D["qpos"]) if TWIN else np.load(sys.argv[2])
horizon = int(sys.argv[3]) if len(sys.argv) > 3 else 120
n = min(int(sys.argv[4]), len(G["qpos"])) if len(sys.argv) > 4 else len(G["qpos"])
threads = int(sys.argv[5]) if len(sys.argv) > 5 else 4
variant = sys.argv[6] if len(sys.argv) > 6 else ""
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
CK = [c for c in G["checkpoints"].tolist() if c <= horizon]


def one(e, variant=variant, chol=0):
    o = Oracle(m, variant=variant)
    if chol:
        o.set_cholesky_order(chol)
    o.set_state(qpos=D["qpos"][e], qvel=D["qvel"][e], warmstart=D["warmstart"][e], pid=D["pid"][e])
    o.set_ctrl(D["ctrl"][e])
    o.set_checkpoints(CK)
    o.move_ee([D["acts"][e][0], D["acts"][e][1], 1.1], 0.05, horizon)
    c = o.get_checkpoints()
    full = np.full((len(CK), m.nq), np.nan)
    full[:len(c)] = c
    return full


with ThreadPoolExecutor(max_workers=threads) as ex:
    T = np.stack(list(ex.map(one, range(n))))
if TWIN:
    with ThreadPoolExecutor(max_workers=threads) as ex:
        gpu = np.stack(list(ex.map(lambda e: one(e, TWIN[1] if len(TWIN) > 1 else "", int(TWIN[2]) if len(TWIN) > 2 else 0), range(n))))
else:
    gpu = G["qpos"][:n, :len(CK)].astype(np.float64).copy()
    gpu[G["steps_taken"][:n, :len(CK)] != np.array(CK)[None, :]] = np.nan
d = np.nanmax(np.abs(gpu - T), axis=2)
first = np.array([next((CK[k] for k in range(len(CK)) if d[e, k] > 1e-6), 2 * CK[-1]) for e in range(n)])
print(json.dumps(dict(scenes=n, first_side=sys.argv[2] if TWIN else "HIP kernel", oracle_variant=variant or "base", checkpoints=CK, median_abs_difference=[float(np.nanmedian(d[:, k])) for k in range(len(CK))],
                      p90_abs_difference=[float(np.nanpercentile(d[:, k], 90)) for k in range(len(CK))],
                      share_within_1e_12=[float(np.nanmean(d[:, k] < 1e-12)) for k in range(len(CK))],
                      first_checkpoint_above_1e_6=dict(median=float(np.median(first)), quartiles=[float(np.percentile(first, 25)), float(np.percentile(first, 75))],
                                                       beyond_horizon=int((first > CK[-1]).sum())))))
if os.environ.get("EARLY_DIV_WHERE"):
    for e in range(min(n, 12)):
        k = min(3, len(CK) - 1)
        dd = np.abs(gpu[e, k] - T[e, k])
        print(e, "step", CK[k], "argmax", int(np.nanargmax(dd)), "max %.1e" % np.nanmax(dd), "robot max %.1e" % np.nanmax(dd[:8]), "objects max %.1e" % np.nanmax(dd[8:]), file=sys.stderr)
