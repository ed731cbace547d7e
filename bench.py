#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json config 2): 4096 batched IT1 scenes per MI355X, physics only, fixed-z grasp attempts
with the 500-step closing check (README.md:20 of the reference). A "step" is one grasp-attempt round over the whole batch
= one launch of the hot path (GraspEnv.step -> move_and_grasp, GraspingEnv.py:62-156,205-386) on every scene.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0. value = env-steps/s of the whole job (2 ms physics steps actually executed, summed over all
scenes and ranks, / max-over-ranks wall time of the K timed rounds); grasp-attempts/s is reported next to it.
Inputs (actions) are resident in HBM before the timed region; rewards stay on the device. N > 1: scenes shard over ranks
(4096 per GPU, weak scaling, seeds keyed by global scene id) and every round ends with ONE RCCL all_gather of the 16-byte
outcome records (mujoco_rl_ur5_amd/sharding.py), inside the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def aimed_actions(qpos, first_id, round_idx, nobj=4):
    """Synthetic input: scene g aims at the settled position of object (g + round) % nobj, z = 0.91."""
    n = qpos.shape[0]
    a = np.zeros((n, 8))
    for e in range(n):
        objs = qpos[e][8:].reshape(-1, 7)
        k = (first_id + e + round_idx) % nobj
        a[e, :3] = [objs[k, 0], -0.6 + objs[k, 1], 0.91]
        a[e, 3] = ((first_id + e) // nobj + round_idx) % 6
    return a


def aimed_actions_rendered(xpos, first_id, round_idx):
    """--workload many / it4 (configs[3] / configs[2] shape): scene g aims at one of the objects lying in the pick bin, 2 cm above its
    centre (what the depth image gives for these object sizes), rotation index cycling through the 6 wrist angles (GraspingEnv.py:40).
    xpos: world positions of the objects [n, nobj, 3]."""
    n = xpos.shape[0]
    a = np.zeros((n, 8))
    for e in range(n):
        objs = xpos[e]
        inbin = np.where((np.abs(objs[:, 0]) < 0.2) & (np.abs(objs[:, 1] + 0.6) < 0.13) & (objs[:, 2] > 0.85))[0]
        k = inbin[(first_id + e + round_idx) % len(inbin)] if len(inbin) else 0
        a[e, :3] = [objs[k, 0], objs[k, 1], objs[k, 2] + 0.02]
        a[e, 3] = (first_id + e + round_idx) % 6
    return a


def cpu_baseline(model, budget_s=10.0, many=False):
    """The fp64 oracle (a port: the reference's own MuJoCo binary cannot exist here) on the host cores, same scenes/actions:
    one scene per native thread on every core (SURVEY.md section 8d ii; oracle/ur5_oracle.cpp ur5o_batch), plus the single-core rate.
    IT1: reset settling + one aimed grasp attempt per scene (= aimed_actions() above); many: the first 100 steps of the drop."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    s1, a1, t1 = O.batch(model, 1, 0.3 * budget_s, 1 if many else 0, 100)
    sn, an, tn = O.batch(model, cores, 0.7 * budget_s, 1 if many else 0, 100)
    what = ("the first 100 steps of the 40-object drop of scenes 0..%d" % (an - 1)) if many else \
           ("IT1 reset settling + one aimed grasp attempt of scenes 0..%d" % (an - 1))
    out = dict(value=sn / tn, unit="env-steps/s", cores=cores, kind="port",
               sample=f"{what} ({sn} physics steps), oracle/ur5_oracle.cpp, one scene per thread on {cores} threads for {tn:.1f} s wall",
               single_core={"value": s1 / t1, "sample": f"{a1} scenes ({s1} steps), {t1:.1f} s on one core"})
    if not many:
        out["grasp_attempts_per_s"] = an / tn
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=None, help="scenes per GPU (default 4096; 2048 for --workload many)")
    ap.add_argument("--workload", choices=("it1", "it4", "many"), default="it1",
                    help="it1 = BASELINE.json configs[1] (the headline metric); it4 = configs[2] shape: in-tree 6-object scene, render + in-tree "
                         "grasp script; many = configs[3] shape: 40-object piles, render + grasp round")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mujoco_rl_ur5_amd import sharding
    from mujoco_rl_ur5_amd.model import load_model
    from mujoco_rl_ur5_amd.native import BatchSim

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    many, rendered = args.workload == "many", args.workload in ("many", "it4")
    model = load_model({"many": "/UR5+gripper/UR5gripper_2_finger_many_objects.xml", "it4": "/UR5+gripper/UR5gripper_2_finger.xml",
                        "it1": "it1_4box"}[args.workload])
    n_local = args.envs if args.envs else (2048 if many else 4096)
    n_total = n_local * world
    lo, hi = sharding.shard_range(n_total, rank, world)
    sim = BatchSim(model, n_local, device_id=local_rank)
    sim.reset(sharding.global_seeds(20, n_total, rank, world), 1, 1000.0)          # GraspingEnv.py:409-477, untimed
    settled = sim.get_state()["qpos"]
    rounds = args.warmup + args.steps
    if rendered:
        xpos = sim.body_xpos()[:, 8:8 + (model.nv - 8) // 6]
        actions = torch.from_numpy(np.stack([aimed_actions_rendered(xpos, lo, r) for r in range(rounds)])).to(dev)
    else:
        actions = torch.from_numpy(np.stack([aimed_actions(settled, lo, r) for r in range(rounds)])).to(dev)   # [rounds, n, 8] f64 in HBM
    if rendered:                                                                  # the observation of every round stays on the device
        img = torch.zeros((n_local, 200, 200, 3), dtype=torch.uint8, device=dev)
        dep = torch.zeros((n_local, 200, 200), dtype=torch.float32, device=dev)
        cam = model.camera_name2id("top_down")
    reward = torch.zeros((rounds, n_local), dtype=torch.int32, device=dev)
    ids = torch.arange(lo, hi, dtype=torch.int32, device=dev)

    def one_round(r):
        if rendered:
            sim.render_dev(img.data_ptr(), dep.data_ptr(), cam, 200, 200, 1)       # get_observation (GraspingEnv.py:390-406), same stream
        sim.grasp_attempt_dev(actions[r].data_ptr(), reward[r].data_ptr(), check_mode=0 if rendered else 1, table_height=0.91)
        sim.sync()                                                             # handle stream -> host; rewards now valid
        rec = torch.stack([ids, torch.zeros_like(ids), actions[r, :, 3].to(torch.int32), reward[r]], dim=1)
        return sharding.gather_outcomes(rec), sim.last_launch_ms()

    for r in range(args.warmup):
        one_round(r)
    c0 = sim.counters()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = []
    for r in range(args.warmup, rounds):
        _, ms = one_round(r)
        kernel_ms.append(ms)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    c1 = sim.counters()
    steps_local = int((c1["total_steps"] - c0["total_steps"]).sum())
    stats = torch.tensor([elapsed, float(steps_local), float(reward[args.warmup:].sum().item())], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed, steps_all, succ_all = float(tmax[0]), float(tot[0]), float(tot[1])
    else:
        steps_all, succ_all = float(steps_local), float(stats[2])
    if rank == 0:
        attempts = args.steps * n_total
        # algorithmic HBM bytes per env-step: (nq + 2 nv + 5 nu + 8) words, read + written, fp64
        words = model.nq + 2 * model.nv + 5 * model.nu + 8
        bytes_per_step = 2 * words * 8
        k_s = sum(kernel_ms) * 1e-3
        achieved = steps_local * bytes_per_step / k_s / 1e9
        # measured HBM bytes per env-step from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/)
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "r01_k_hbm_traffic.json")
        if os.path.exists(tp) and not rendered:
            with open(tp) as f:
                tj = json.load(f)
            traffic = tj["hbm_bytes_per_env_step"] * steps_local / args.steps
            traffic_src = "profiles/r01_k_hbm_traffic.json: (FETCH_SIZE + WRITE_SIZE) per env-step x env-steps of an average timed launch"
        out = {
            "metric": f"env-steps/sec (+ grasp-attempts/sec), {n_local} parallel UR5 scenes per MI355X",
            "value": steps_all / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "grasp_attempts_per_s": attempts / elapsed, "grasp_success_rate": succ_all / attempts,
            "env_steps_per_attempt": steps_all / attempts,
            "newton_iters_per_step": float((c1["solver_iters"] - c0["solver_iters"]).sum()) / max(1, steps_local),
            "status_bits": int(np.bitwise_or.reduce(c1["status"])),
            "config": {"workload": ("BASELINE.json configs[3] shape: IT5 many-object piles (UR5gripper_2_finger_many_objects.xml, 40 objects, "
                                    "condim 6), 200x200 RGB-D render + multi-discrete rotation action + in-tree grasp script per step") if many else
                                   ("BASELINE.json configs[2] shape: IT4 (in-tree UR5gripper_2_finger.xml, 3 boxes + 3 spheres), 200x200 RGB-D render + "
                                    "grasp height from the object top + in-tree grasp script (closing check at the drop position) per step") if rendered else
                                   ("BASELINE.json configs[1]: IT1 (UR5gripper_2_finger.xml robot + bins, 4 equal 4 cm boxes), physics only, "
                                    "fixed z = 0.91, lift + 500-step closing check, one grasp-attempt round per step"),
                       "scenes_per_gpu": n_local, "scenes_total": n_total, "solver": "Newton (MuJoCo default), tol 1e-10",
                       "timestep_s": model.opt["timestep"], "parallelism": f"scenes sharded x{world}, 1 all_gather of 16 B outcome records per round"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "ur5m_run_kernel<248>" if many else ("ur5_run_kernel<44>" if rendered else "ur5_run_kernel<32>"), "bytes_per_env_step": bytes_per_step,
                         "avg_launch_ms": float(np.mean(kernel_ms)), "env_steps_per_launch": steps_local / args.steps,
                         "note": "algorithmic state bytes x env-steps / HIP-event kernel time on the handle's stream (rank 0). The kernel keeps a "
                                 "scene in LDS for a whole grasp attempt, so real HBM traffic is far below the algorithmic figure; the step is "
                                 "latency/VALU bound (DESIGN.md)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, many=many)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
