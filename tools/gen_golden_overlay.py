#!/usr/bin/env python3
"""Read the pick-bin frame off the reference's media/overlay.png (README "Grasp chance overlay": a 200x200 top-down observation shown
by matplotlib WITH pixel axes) into tests/golden/overlay_png.json. Run in the build container (reads /root/reference).

The figure's tick marks give the data-pixel scale (1.851 figure px per image px); the frame of the pick bin (wall tops, brighter than the
floor around them) is located on three scan lines per side as the strongest brightness step. Resolution: 0.54 image px per figure px.
The picture is of an older revision of the scene (wall tops at z = 0.86, plate at 0.89: the variant UR5gripper_2_finger.xml:116-123 keeps
as a comment; six objects) -- the frame's outline is 0.66 m x 0.52 m in both."""
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
g = np.array(Image.open(os.path.join(REF, "media", "overlay.png")).convert("RGB")).astype(float).sum(2) / 3
# tick marks just outside the axes frame: x ticks (0..175 step 25) in row 406, y ticks in column 42
xt = [x for x in range(40, 425) if g[406, x] < 120]
yt = [y for y in range(30, 412) if g[y, 42] < 120]
assert len(xt) == 8 and len(yt) == 8, (xt, yt)
sx, x0 = np.polyfit(np.arange(8) * 25.0, xt, 1)
sy, y0 = np.polyfit(np.arange(8) * 25.0, yt, 1)


def edge(profile, lo, hi, sign):
    d = np.diff(profile[lo:hi]) * sign
    i = int(np.argmax(d))
    return lo + i + 0.5


rows = [int(round(y0 + v * sy)) for v in (60, 100, 140)]
cols = [int(round(x0 + v * sx)) for v in (40, 100, 160)]
left = [(edge(g[r], int(x0 + 15 * sx), int(x0 + 45 * sx), +1) - x0) / sx for r in rows]
right = [(edge(g[r], int(x0 + 164 * sx), int(x0 + 176 * sx), -1) - x0) / sx for r in rows]
top = [(edge(g[:, c], int(y0 + 36 * sy), int(y0 + 50 * sy), +1) - y0) / sy for c in cols]
bottom = [(edge(g[:, c], int(y0 + 145 * sy), int(y0 + 165 * sy), -1) - y0) / sy for c in cols]
out = dict(source="media/overlay.png", figure_px_per_image_px=[float(sx), float(sy)],
           frame_edges_image_px=dict(left=float(np.mean(left)), right=float(np.mean(right)), top=float(np.mean(top)), bottom=float(np.mean(bottom))),
           spread=dict(left=float(np.ptp(left)), right=float(np.ptp(right)), top=float(np.ptp(top)), bottom=float(np.ptp(bottom))),
           note="edges in imshow coordinates (pixel k is centred on k); frame = outer outline of the four wall tops of pick_box, "
                "0.66 m x 0.52 m centred under the camera, at z = 0.86 in the pictured revision (0.88 in the shipped file)")
# coarse layout: a 40 x 40 grid (cells of 5 image px); a cell is "structure" when its colour is not the bluish floor (B - R < 6). Pins the
# image orientation incl. the left-right mirror of get_image_data (:708-727) and where pedestal, bins and frame sit in the picture.
im = np.array(Image.open(os.path.join(REF, "media", "overlay.png")).convert("RGB")).astype(float)
mask = []
for i in range(40):
    row = ""
    for j in range(40):
        px, py = int(round(x0 + (5 * j + 2) * sx)), int(round(y0 + (5 * i + 2) * sy))
        p = im[py - 2:py + 3, px - 2:px + 3].reshape(-1, 3).mean(0)
        row += "#" if p[2] - p[0] < 6 else "."
    mask.append(row)
out["structure_mask_40x40"] = mask
print(json.dumps({k: v for k, v in out.items() if k != "structure_mask_40x40"}, indent=1))
print("\n".join(mask))
with open(os.path.join(ROOT, "tests", "golden", "overlay_png.json"), "w") as f:
    json.dump(out, f, indent=1)
