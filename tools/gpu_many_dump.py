#!/usr/bin/env python3
"""GPU side of the pile statistics of round 4 (run through gpurun; the CPU side is tools/pile_chaos_floor.py, run wherever there are cores):
settle a pool of 40-object piles on the HIP many-object kernel, aim with tools/pile_aim.py, take the `keep` best-scoring scenes, run their grasp attempt
TWICE from the same device records (run-to-run determinism of ur5m_run_kernel: every word of both results must be equal), and write states + results to
an .npz that the oracle replays.     python tools/gpu_many_dump.py [pool=3072] [keep=256] [out=gpurun_out/r04_many_states.npz] [seed0=7000]"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from pile_aim import pick_box

pool = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/r04_many_states.npz"
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 7000
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
sim = BatchSim(m, pool)
seeds = seed0 + np.arange(pool, dtype=np.uint64)
sim.reset(seeds, 1, 1000.0)
settle_ms = sim.last_launch_ms()
twin = BatchSim(m, pool)                                      # the settle itself, twice (a fresh handle: a second reset of the same one starts from another PID history)
twin.reset(seeds, 1, 1000.0)
torch.cuda.synchronize()
settle_equal = bool(torch.equal(sim.state_tensor("cuda"), twin.state_tensor("cuda")))
twin.close()
st, ctrl = sim.get_state(), sim.get_ctrl()
acts, rots, scores = np.zeros((pool, 3)), np.zeros(pool, dtype=np.int64), np.full(pool, np.nan)
acts[:] = [0.0, -0.6, 1.0]
for e in range(pool):
    b = pick_box(m, st["qpos"][e])
    if b is not None:
        acts[e], rots[e], scores[e] = b[1], b[2], b[3]
sel = np.argsort(np.where(np.isnan(scores), 1e9, scores), kind="stable")[:keep]
rec0 = sim.state_tensor("cuda").clone()
runs = []
for k in range(2):
    sim.state_tensor("cuda").copy_(rec0)
    torch.cuda.synchronize()
    rew, ps, pr = sim.grasp_attempt(acts, rot=rots, check_mode=0)
    ms = sim.last_launch_ms()
    torch.cuda.synchronize()
    runs.append((rew.copy(), ps.copy(), pr.copy(), sim.state_tensor("cuda").clone().cpu().numpy(), ms))
a, b = runs
identical = dict(reward=bool(np.array_equal(a[0], b[0])), phase_steps=bool(np.array_equal(a[1], b[1])), phase_result=bool(np.array_equal(a[2], b[2])),
                 records=bool(np.array_equal(a[3].view(np.uint64), b[3].view(np.uint64))),
                 scenes_with_any_difference=int((np.any(a[3].view(np.uint64) != b[3].view(np.uint64), axis=1) | (a[0] != b[0])).sum()))
c = sim.counters()
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
np.savez_compressed(out, sel=sel, seeds=seeds[sel], qpos=st["qpos"][sel], qvel=st["qvel"][sel], warmstart=st["warmstart"][sel], pid=st["pid"][sel], ctrl=ctrl[sel],
                    acts=acts[sel], rots=rots[sel], scores=scores[sel], gpu_reward=a[0][sel], gpu_phase_steps=a[1][sel], gpu_phase_result=a[2][sel],
                    gpu_qpos_after=a[3][sel][:, :m.nq], ncon_max=c["ncon_max"], solver_iters=c["solver_iters"], total_steps=c["total_steps"])
print(json.dumps(dict(pool=pool, kept=int(len(sel)), worst_kept_score=float(np.nanmax(scores[sel])), settle_kernel_ms=settle_ms, settle_twice_bit_identical=settle_equal,
                      grasp_kernel_ms=[a[4], b[4]], grasp_twice_bit_identical=identical, gpu_positives_kept=int(a[0][sel].sum()), gpu_success_pool=float(a[0].mean()),
                      status_nonzero=int((c["status"] != 0).sum()), ncon_max_hist=np.bincount(np.minimum(c["ncon_max"], 160) // 10, minlength=17).tolist(),
                      ncon_max_max=int(c["ncon_max"].max()), newton_iters_per_step=float(c["solver_iters"].sum() / max(1, c["total_steps"].sum())))))
