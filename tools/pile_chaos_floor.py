#!/usr/bin/env python3
"""How chaotic IS a grasp attempt on a 40-object pile?  (CPU; reads the states tools/gpu_many_dump.py wrote.)

The oracle replays every kept scene three times from the kernel's settled state: as is; with the contact list of every step REVERSED (the same step
mathematically -- only the association order of the sums over contacts changes, a last-bit perturbation); and with one object's x moved by 1 ulp
before the first step. Agreement of the reward bit / result codes / step counts between the oracle and its own perturbed twins is the floor that
GPU-vs-oracle agreement (two texts of one algorithm, different summation orders) can be measured against: GraspingEnv.py:327 is the bit,
MujocoController.py:379 the single deterministic thread the reference runs.
      python tools/pile_chaos_floor.py gpurun_out/r04_many_states.npz [threads=8] [limit=256] > profiles/r04_pile_chaos_floor.json"""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from oracle.oracle import Oracle

src = sys.argv[1]
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 8)
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
D = {k: v for k, v in np.load(src).items()}                  # in memory: the threads index these arrays
n = min(limit, len(D["sel"]))
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
VARIANTS = tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else ("base", "reversed_contacts", "one_ulp")   # "base" alone: GPU-vs-oracle of a new kernel against the recorded floor


def one(job):
    e, variant = job
    o = Oracle(m, variant="fmadyn" if variant == "fmadyn" else "")   # round 6: the control twins of tools/pile_divergence_time.py -- the oracle's own text on another arithmetic
    q = D["qpos"][e].copy()
    if variant == "one_ulp":
        q[8] = np.nextafter(q[8], np.inf)                     # x of object 0
    o.set_state(qpos=q, qvel=D["qvel"][e], warmstart=D["warmstart"][e], pid=D["pid"][e])
    o.set_ctrl(D["ctrl"][e])
    if variant == "reversed_contacts":
        o.set_contact_order(1)
    if variant == "rsqrt_cholesky":
        o.set_cholesky_order(2)
    r, ps, pr = o.grasp_attempt(D["acts"][e], int(D["rots"][e]), 0)
    return int(r), ps.copy(), pr.copy(), o.get_state()["qpos"], o.solver_iters, o.total_steps


t0 = time.time()
jobs = [(e, v) for e in range(n) for v in VARIANTS]
with ThreadPoolExecutor(max_workers=threads) as ex:
    res = list(ex.map(one, jobs))
R = {v: [res[e * len(VARIANTS) + k] for e in range(n)] for k, v in enumerate(VARIANTS)}


def compare(A, B):
    ra, rb = np.array([a[0] for a in A]), np.array([b[0] for b in B])
    codes = np.mean([a[2].tolist() == b[2].tolist() for a, b in zip(A, B)])
    steps = np.mean([a[1].tolist() == b[1].tolist() for a, b in zip(A, B)])
    closed = np.mean([(a[1][5] > 300) == (b[1][5] > 300) for a, b in zip(A, B)])
    arm = [float(np.abs(a[3][:8] - b[3][:8]).max()) for a, b in zip(A, B)]
    objs = [float(np.abs(a[3][8:] - b[3][8:]).max()) for a, b in zip(A, B)]
    return dict(grasp_bit_agreement=float((ra == rb).mean()), positives=[int(ra.sum()), int(rb.sum())],
                second_positive_where_first_positive=float(rb[ra == 1].mean()) if ra.any() else None,
                second_negative_where_first_negative=float((1 - rb[ra == 0]).mean()) if (ra == 0).any() else None,
                phase_result_codes_identical=float(codes), phase_steps_identical=float(steps), closed_on_something_agreement=float(closed),
                arm_abs_difference_median=float(np.median(arm)), object_qpos_difference_median=float(np.median(objs)),
                scenes_bit_identical_final_state=int(sum(o == 0.0 and a == 0.0 for o, a in zip(objs, arm))))


gpu = [(int(D["gpu_reward"][e]), D["gpu_phase_steps"][e], D["gpu_phase_result"][e], D["gpu_qpos_after"][e]) for e in range(n)]
if VARIANTS == ("base",):
    print(json.dumps(dict(scenes=n, source=os.path.basename(src), oracle_seconds=round(time.time() - t0, 1), threads=threads, gpu_vs_oracle=compare(R["base"], gpu),
                          note="GPU kernel vs the oracle from the kernel's settled states; the floor to compare with is profiles/r04_pile_chaos_floor_256of3072.json")))
    sys.exit(0)
if VARIANTS != ("base", "reversed_contacts", "one_ulp"):   # any other list (round 6: base,fmadyn,rsqrt_cholesky): every pair of variants, and the GPU against each
    out = dict(scenes=n, source=os.path.basename(src), oracle_seconds=round(time.time() - t0, 1), threads=threads, variants=list(VARIANTS))
    for i, a in enumerate(VARIANTS):
        for b in VARIANTS[i + 1:]:
            out["oracle_%s_vs_oracle_%s" % (a, b)] = compare(R[a], R[b])
        out["gpu_vs_oracle_%s" % a] = compare(R[a], gpu)
    out["note"] = "all runs start from the HIP kernel's settled state of the same scenes; fmadyn = fused dynamics + strict geometry (the pile unit's arithmetic split), rsqrt_cholesky = pivots by reciprocal square root"
    print(json.dumps(out))
    sys.exit(0)
out = dict(scenes=n, source=os.path.basename(src), oracle_seconds=round(time.time() - t0, 1), threads=threads,
           oracle_newton_iters_per_step=float(sum(r[4] for r in R["base"]) / max(1, sum(r[5] for r in R["base"]))),
           oracle_vs_oracle_reversed_contacts=compare(R["base"], R["reversed_contacts"]),
           oracle_vs_oracle_one_ulp=compare(R["base"], R["one_ulp"]),
           oracle_reversed_vs_oracle_one_ulp=compare(R["reversed_contacts"], R["one_ulp"]),
           gpu_vs_oracle=compare(R["base"], gpu), gpu_vs_oracle_reversed_contacts=compare(R["reversed_contacts"], gpu), gpu_vs_oracle_one_ulp=compare(R["one_ulp"], gpu),
           note="all runs start from the HIP kernel's settled state of the same scenes; 'floor' = agreement of the oracle with its own rounding-level twins")
fl = [out[k]["grasp_bit_agreement"] for k in ("oracle_vs_oracle_reversed_contacts", "oracle_vs_oracle_one_ulp", "oracle_reversed_vs_oracle_one_ulp")]
gp = [out[k]["grasp_bit_agreement"] for k in ("gpu_vs_oracle", "gpu_vs_oracle_reversed_contacts", "gpu_vs_oracle_one_ulp")]
out["summary"] = dict(floor_grasp_bit_agreement_min_mean_max=[min(fl), sum(fl) / 3, max(fl)], gpu_grasp_bit_agreement_min_mean_max=[min(gp), sum(gp) / 3, max(gp)],
                      gpu_at_or_above_floor_minus_1_percent=bool(min(gp) >= min(fl) - 0.01))
print(json.dumps(out))
