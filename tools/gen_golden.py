#!/usr/bin/env python3
"""Generate the small committed fixtures under tests/golden/ from the reference tree (run HERE, not on the GPU box).

  urdf_chain.json   joint origins/axes of the ikpy chain base_link -> ee_link (ur5_gripper.urdf:61-234)
  console_png.json  the one recorded input/output pair of the reference (media/console.png)
  oracle_grasp.json oracle trajectories on the IT1 scene (seeded), so the GPU box can detect oracle drift
"""
import json, os, sys
import xml.etree.ElementTree as ET
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

urdf = ET.parse(os.path.join(REF, "UR5+gripper", "ur5_gripper.urdf")).getroot()
joints = {j.find("parent").get("link"): j for j in urdf.findall("joint") if j.find("parent") is not None and j.get("type") in ("revolute", "fixed")}
chain, link = [], "base_link"
by_parent = {}
for j in urdf.findall("joint"):
    if j.find("parent") is not None:
        by_parent.setdefault(j.find("parent").get("link"), []).append(j)
while link != "ee_link":
    j = by_parent[link][0]           # ikpy follows the first child joint
    o = j.find("origin")
    chain.append(dict(name=j.get("name"), type=j.get("type"), xyz=[float(x) for x in o.get("xyz").split()],
                      rpy=[float(x) for x in o.get("rpy").split()],
                      axis=[float(x) for x in j.find("axis").get("xyz").split()] if j.find("axis") is not None else [0, 0, 1]))
    link = j.find("child").get("link")
json.dump(chain, open(os.path.join(OUT, "urdf_chain.json"), "w"), indent=1)

# media/console.png: "Action: Pixel X: 136, Pixel Y: 80 ... Transformed into world coordinates: [-0.16551974 -0.50804459]", z = 0.88999999
json.dump(dict(pixel=[136, 80], world=[-0.16551974, -0.50804459, 0.88999999], phase_steps=[362, 136, 202, 631, 33]),
          open(os.path.join(OUT, "console_png.json"), "w"), indent=1)

from mujoco_rl_ur5_amd.model import load_model
from oracle.oracle import Oracle
m = load_model("it1_4box")
rec = []
for seed in (20, 21, 22):
    o = Oracle(m); o.reset(seed, 1, True)
    q0 = o.get_state()["qpos"].copy()
    objs = q0[8:].reshape(-1, 7); k = (seed - 20) % 4
    act = [float(objs[k, 0]), float(-0.6 + objs[k, 1]), 0.91]
    r, ps, pr = o.grasp_attempt(act, rot=(seed - 20) % 6, check_mode=0)
    rec.append(dict(seed=seed, settled_qpos=q0.tolist(), action=act, rot=(seed - 20) % 6, reward=int(r), phase_steps=ps.tolist(),
                    phase_result=pr.tolist(), final_qpos=o.get_state()["qpos"].tolist(), total_steps=int(o.total_steps)))
json.dump(rec, open(os.path.join(OUT, "oracle_grasp.json"), "w"))
print("wrote", os.listdir(OUT))
