#!/usr/bin/env python3
"""Digitise the reference's recorded arm trajectory, media/plot_1.png (README.md:120-123), into tests/golden/plot_1.json.

The figure is the output of MJ_Controller.move_group_to_joint_target(group="Arm", plot=True) (MujocoController.py:303-304,338-339,
639-705): one subplot per arm joint, joint angle [rad] over controller steps, the target as a green dashed line and target +- tolerance
in red. It is the only time series of simulator state the reference repository holds. Run in the build container (reads
/root/reference); the GPU box only sees the committed JSON.

What is read off the pixels (900 x 600 px, so one pixel is 1.85 steps / 7-9 mrad):
  * axis calibration from the tick marks (least-squares line through all ticks of an axis; residual < 2.5 mrad),
  * x limits 0..440 with matplotlib's 5 % margins => the samples span steps 20..420 every 20 steps (the plotted code version sampled
    every 20th step; today's samples every 2nd, :338),
  * the curve's y at each sample step (colour-weighted centroid of the blue line in the two neighbouring pixel columns),
  * target and tolerance from the dashed lines.
"""
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
im = np.array(Image.open(os.path.join(REF, "media", "plot_1.png")).convert("RGB")).astype(float)
ink = 255 - im.sum(2) / 3
R, G, B = im[:, :, 0], im[:, :, 1], im[:, :, 2]
blue = np.clip((B - R) / 150.0, 0, 1) * ((B > G) & (G > R))
green = ((G > R + 40) & (G > B + 40)).astype(float)
red = ((R > G + 80) & (R > B + 80)).astype(float)


def centroids(profile, lo, thr=30):
    out, cur = [], []
    for i, v in enumerate(list(profile) + [0]):
        if v > thr:
            cur.append((i + lo, v))
        elif cur:
            out.append(sum(p * w for p, w in cur) / sum(w for p, w in cur))
            cur = []
    return out


# axes frames in pixels (left, right, top, bottom) and the tick labels (top value, bottom value, count) as printed in the figure
AXES = {"shoulder_pan_joint": (45, 283, 30, 255, (0.0, -2.0, 9)), "shoulder_lift_joint": (331, 569, 30, 255, (-0.2, -1.2, 6)),
        "elbow_joint": (617, 855, 30, 255, (1.6, 0.2, 8)), "wrist_1_joint": (45, 283, 345, 570, (0.0, -2.0, 9)),
        "wrist_2_joint": (331, 569, 345, 570, (-0.2, -1.6, 8)), "wrist_3_joint": (617, 855, 345, 570, (0.04, -0.04, 5))}
PX_PER_STEP = 27.05 / 50.0          # x ticks every 50 steps are 27.05 px apart; step 0 sits on the left frame line
out = {"source": "media/plot_1.png", "sample_steps": list(range(20, 421, 20)), "joints": {}}
for name, (xl, xr, yt, yb, (vt, vb, n)) in AXES.items():
    ty = centroids(ink[yt - 2:yb + 3, xl - 3:xl - 1].mean(1), yt - 2)
    ty = [t for t in ty if yt + 1 < t < yb - 1] if len(ty) != n else ty
    if len(ty) == n - 1:            # a tick that coincides with the frame's top line (value vt) merges with it
        ty = [float(yt) + 1.0] + ty
    assert len(ty) == n, (name, ty)
    vals = np.linspace(vt, vb, n)
    fit = np.polyfit(ty[1:] if len(ty) > 3 else ty, vals[1:] if len(ty) > 3 else vals, 1)
    resid = float(np.abs(np.polyval(fit, ty[1:]) - vals[1:]).max())
    samples = []
    for s in out["sample_steps"]:
        xp = xl + s * PX_PER_STEP
        x0 = int(np.floor(xp))
        f = xp - x0
        acc = []
        for xx, w in ((x0, 1 - f), (x0 + 1, f)):
            col = blue[yt + 1:yb, xx]
            if col.sum() > 0.3:
                acc.append(((np.arange(yt + 1, yb) * col).sum() / col.sum(), w))
        samples.append(float(np.polyval(fit, sum(a * b for a, b in acc) / sum(b for a, b in acc))) if acc else None)
    gy = [y + yt + 1 for y, v in enumerate(green[yt + 1:yb, xl + 5:xr - 5].sum(1)) if v > 40]
    ry = [y + yt + 1 for y, v in enumerate(red[yt + 1:yb, xl + 5:xr - 5].sum(1)) if v > 40]
    target = float(np.polyval(fit, np.mean(gy)))
    band = [float(np.polyval(fit, y)) for y in ry]
    tol = (max(band) - min(band)) / 2
    out["joints"][name] = {"target": round(target, 4), "tolerance_read": round(tol, 4), "rad_per_px": round(abs(float(fit[0])), 5),
                           "tick_fit_residual": round(resid, 5), "q": [None if v is None else round(v, 4) for v in samples]}
    print(name, "target", round(target, 3), "tol", round(tol, 3), "rad/px", round(abs(float(fit[0])), 4), "resid", round(resid, 4))
    print("   ", [None if v is None else round(v, 3) for v in samples])
out["note"] = ("Samples the blue line hides behind a dashed line are null. The trajectory ends between step 420 (last sample) and 440 (next "
               "sample that is absent). Start state: arm at qpos 0 (curves extrapolate to 0 at step 0).")
with open(os.path.join(ROOT, "tests", "golden", "plot_1.json"), "w") as f:
    json.dump(out, f, indent=1)
