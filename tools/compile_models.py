#!/usr/bin/env python3
"""Compile the reference's MJCF scenes into the shipped ``assets/*.ur5m`` blobs.

The GPU box has no /root/reference, so the derived model constants (hull vertices, inertias, pair
lists -- data, not source) are generated HERE and committed. Re-run after touching mjcf.py:

    python tools/compile_models.py [/root/reference]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.mjcf import compile_mjcf  # noqa: E402
from mujoco_rl_ur5_amd.model import ASSET_DIR  # noqa: E402

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = os.path.join(ref, "UR5+gripper")

# "IT1, 4 equal objects" -- 4 boxes, half-size 0.02, slide x3 + ball
# joints as UR5gripper_2_finger.xml:233-239; start heights staggered so that they never overlap.
IT1_OBJECTS = [dict(name=f"box_{k + 1}", type="box", size=[0.02, 0.02, 0.02], pos=[0.0, -0.6, 0.95 + 0.1 * k],
                    joints="slide3ball", rgba=c)
               for k, c in enumerate([(0.72, 0.52, 0.32, 1), (0.0, 0.5, 0.8, 1), (0.8, 0.8, 0.1, 1), (0.9, 0.2, 0.2, 1)])]

jobs = [("ur5_2f.ur5m", "UR5gripper_2_finger.xml", None, False),
        ("ur5_2f_it1_4box.ur5m", "UR5gripper_2_finger.xml", IT1_OBJECTS, False),
        ("ur5_2f_many.ur5m", "UR5gripper_2_finger_many_objects.xml", None, False),
        # the same pile scene WITH the seven arm-link hulls colliding, as in the reference (UR5gripper_2_finger_many_objects.xml:158-185,
        # contype 1; the arm hulls are capped at 32 vertices, the gripper's are full): the many-object engine has a contact slot for every robot weld group
        ("ur5_2f_many_armcol.ur5m", "UR5gripper_2_finger_many_objects.xml", None, True)]
os.makedirs(ASSET_DIR, exist_ok=True)
for out, xml, objs, arm in jobs:
    m = compile_mjcf(os.path.join(src, xml), objects=objs, arm_collision=arm)
    m.save(os.path.join(ASSET_DIR, out))
    print(f"{out}: nq={m.nq} nv={m.nv} nu={m.nu} nbody={m.nbody} ngeom={m.ngeom} npair={len(m.pair_geom1)} "
          f"ntree={m.ntree} hullverts={len(m.mesh_vert)} bytes={os.path.getsize(os.path.join(ASSET_DIR, out))}")
