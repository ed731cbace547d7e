#!/usr/bin/env python3
"""Are the engine's contacts BIT-EQUAL to the oracle's from the same state?  (round-5 verdict 1b; CPU only: the lane-emulation build of the engine source against
oracle/libur5_oracle.so, both compiled without fused multiply-adds -- the arithmetic of the HIP pile unit's geometry, ur5_engine.h UR5_STRICT.)

Settled 40-object piles (oracle reset + settle), then `steps` steps along the oracle's trajectory: before each the engine is put into the oracle's state, both run
one forward pass, and every contact (distance, position, normal) is compared word for word, keyed by (geom pair, number within the pair). Reported per geom-type
pair: contacts, contacts with any differing bit, worst absolute difference; plus body poses (the kinematics' share).
    python tools/contact_bits.py [piles=16] [settle_steps=400] [steps=6] [lib=<lane emulation> | gpu]"""
import json, os, sys
from collections import defaultdict
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from oracle.oracle import Oracle

TYPES = {0: "plane", 2: "sphere", 3: "capsule", 5: "cylinder", 6: "box", 7: "mesh"}


def compare(m, sim, o, acc, worst):
    """one forward pass of both from the oracle's state; returns (contacts, differing contacts)"""
    st = o.get_state()
    sim.set_state(qpos=st["qpos"][None], qvel=st["qvel"][None], warmstart=st["warmstart"][None], pid=st["pid"][None])
    o.forward()
    d = sim.forward_debug()
    oc = o.contacts()
    n = int(d["ncon"][0])
    ec = d["contacts"][0][:n]
    # both lists are in geom-pair order (the engine numbers only the collidable geoms: its ids are not the oracle's); equal counts are paired by position
    tot = bad = 0
    gt = m.geom_type
    if n == len(oc):
        for e, c in zip(ec, oc):
            name = "%s-%s" % (TYPES.get(int(gt[int(c[7])]), "?"), TYPES.get(int(gt[int(c[8])]), "?"))
            a = acc[name]
            a[0] += 1; tot += 1
            if not np.array_equal(e[:7].view(np.uint64), c[:7].view(np.uint64)):
                a[1] += 1; bad += 1
                worst[name] = max(worst[name], float(np.abs(e[:7] - c[:7]).max()))
                if os.environ.get("CONTACT_BITS_VERBOSE"):
                    print(name, int(c[7]), int(c[8]), "dist %.3e pos %.3e normal %.3e" % (abs(e[0] - c[0]), np.abs(e[1:4] - c[1:4]).max(), np.abs(e[4:7] - c[4:7]).max()), file=sys.stderr)
    acc["_count_mismatch"][0] += int(n != len(oc))
    return tot, bad


if __name__ == "__main__":
    piles = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    settle = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    if len(sys.argv) > 4:
        lib = None if sys.argv[4] == "gpu" else sys.argv[4]                      # "gpu": the shipped libur5sim.so on the MI355X
    else:
        import conftest
        lib = conftest.build_emul()
    m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
    sim = BatchSim(m, 1, lib_path=lib)
    acc, worst = defaultdict(lambda: [0, 0, 0]), defaultdict(float)
    tot = bad = 0
    scene_steps_with_a_difference = 0
    for p in range(piles):
        o = Oracle(m)
        o.reset(7000 + p, 1, False)
        o.step(settle)
        for k in range(steps):
            t, b = compare(m, sim, o, acc, worst)
            tot += t; bad += b
            scene_steps_with_a_difference += int(b > 0)
            o.step(1)
    print(json.dumps(dict(piles=piles, settle_steps=settle, steps=steps, contacts=tot, contacts_with_a_differing_bit=bad, share=bad / max(1, tot),
                          scene_steps=piles * steps, scene_steps_with_a_difference=scene_steps_with_a_difference,
                          by_pair_type={k: dict(contacts=v[0], differing=v[1], missing_in_engine=v[2], worst_abs=worst.get(k, 0.0)) for k, v in sorted(acc.items())})))
