#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json config 2): 4096 batched IT1 scenes, physics only, fixed-z grasp attempts with the 500-step
closing check (README.md:20 of the reference). A "step" is one grasp-attempt round over the whole batch = the hot path
(GraspEnv.step -> move_and_grasp, GraspingEnv.py:62-156,205-386, plus GraspEnv.reset_model at episode boundaries, :409-477).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (stationary by construction, stated in the JSON line as config.rule):
  * every scene lives through episodes of EP = 4 rounds: one grasp attempt per round, and after the last one reset_model (objects
    re-sampled from the scene's SplitMix64 stream, arm to home, 1000 ms settle). Scene g ends an episode in the rounds r with
    (r + 1 + g) % 4 == 0, so in every round exactly a quarter of the batch resets: the mix of settling / full-plate / nearly-empty-plate
    scenes is the same in every round, warm-up or timed. Attempt and reset of a scene run in ONE launch (ur5_grasp_attempt_reset_dev).
  * rule "aimed" (headline): in round j of its episode scene g aims at the CURRENT position of the first of boxes (g + j + i) % 4,
    i = 0.., that still lies on the pick plate (read from the engine's state records on the device), z = 0.91, rotation cycling;
    an empty plate gets an attempt at its centre.
  * rule "uniform" (second figure, SURVEY.md section 8d's own rule): a uniformly drawn pixel of the 200x200 top-down image whose
    back-projection lies on the table (the agent's rejection rule, Grasping_Agent_multidiscrete.py:266-280), rotation 0, z = 0.91.
Everything a round needs (seeds, action records, dispatch order) is produced on the device on the SAME HIP stream as the engine's
kernel (ur5_set_stream): no host synchronisation inside a round except the outcome all_gather's own. A launch ends with its slowest
scene, so scenes are dispatched longest-expected-first (ur5_set_order_dev): episode-ending scenes (attempt + 500 settle steps), then
scenes with a box to carry, then attempts on an empty plate. The order changes the makespan only, never a result.
For the same reason the rank's scenes are simulated as --groups G = 2 scene groups (one engine handle + one HIP stream each, half of the
scenes each): the next round of one group is queued behind its current one while the other group's round is still running, so the wave
slots that the tail of a launch leaves empty are taken by the other group's launch. Scenes, seeds and results are those of one group.

Prints ONE JSON line on rank 0. value = env-steps/s of the whole job (2 ms physics steps actually executed, summed over all scenes
and ranks, / max-over-ranks wall time of the K timed rounds); grasp-attempts/s next to it. N > 1: scenes shard over ranks
(--scaling weak: 4096 per GPU; strong: 4096 in total), seeds keyed by global scene id, and every round ends
with ONE RCCL all_gather of the 16-byte outcome records (mujoco_rl_ur5_amd/sharding.py), inside the timed region.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
EP = 4          # rounds per episode
BASE_SEED = 20  # Grasping_Agent_multidiscrete.py:64


class It1Rounds:
    """Device-side driver of the stationary IT1 workload for one rank: flags / seeds of the scenes that reset, action records."""

    def __init__(self, torch, model, sim, dev, lo, n_local, n_total, rule):
        from mujoco_rl_ur5_amd.controller import MJ_Controller
        self.torch, self.sim, self.rule, self.n, self.n_total = torch, sim, rule, n_local, n_total
        self.gid = torch.arange(lo, lo + n_local, dtype=torch.int64, device=dev)
        self.state = sim.state_tensor(dev)                                        # [n, 192] f64, aliases the engine's records
        # objects: 3 slides + ball each (UR5gripper_2_finger.xml:233-239): world position = body_pos + slide offsets
        self.nobj = (model.nq - 8) // 7
        obj_bodies = [int(model.jnt_bodyid[list(model.jnt_qposadr).index(8 + 7 * k)]) for k in range(self.nobj)]
        self.pos0 = torch.from_numpy(np.asarray(model.body_pos)[obj_bodies].copy()).to(dev)   # [nobj, 3]
        # pinhole of the top-down camera at table height (MujocoController.py:742-806): world (x, y) <-> pixel is affine at fixed z
        ctl = MJ_Controller(model, sim, None)
        z_t = 0.91
        cam_z = float(model.cam_pos0[model.camera_name2id("top_down")][2])
        p00 = ctl.pixel_2_world(0, 0, cam_z - z_t, 200, 200)
        p10 = ctl.pixel_2_world(1, 0, cam_z - z_t, 200, 200)
        p01 = ctl.pixel_2_world(0, 1, cam_z - z_t, 200, 200)
        self.px0 = (float(p00[0]), float(p00[1]))
        self.dxdpx, self.dydpy = float(p10[0] - p00[0]), float(p01[1] - p00[1])
        assert abs(p10[1] - p00[1]) < 1e-9 and abs(p01[0] - p00[0]) < 1e-9, "top-down camera: pixel axes are world axes"
        if rule == "uniform":                                                    # pixels whose back-projection lies on the plate
            px = np.arange(200)
            X = self.px0[0] + self.dxdpx * px
            Y = self.px0[1] + self.dydpy * px
            ok = (np.abs(X)[None, :] <= 0.27) & (np.abs(Y + 0.6)[:, None] <= 0.19)
            self.table_pixels = torch.from_numpy(np.flatnonzero(ok.ravel())).to(dev)       # flat index = y * 200 + x
            self.gen = torch.Generator(device=dev).manual_seed(BASE_SEED)

    def reset_seeds(self, r):
        """int64 [n] (read as uint64 by the engine): the seed of the episode that scene g starts after round r's attempt when
        (r + 1 + g) % EP == 0, else 0 = no reset. Episode 0 is the untimed reset before the first round."""
        k = self.gid + r + 1
        seeds = BASE_SEED + self.gid + self.n_total * (k // EP)                  # == sharding.global_seeds(20, n_total, ..., episode)
        return self.torch.where((k % EP) == 0, seeds, self.torch.zeros_like(seeds))

    def launch(self, r, reward_row, check_mode=1):
        """One round on the engine's stream: action records from the current state, longest-expected-first dispatch order, ONE launch of
        attempt (+ episode reset). Returns (action records [n, 8], aimed pixel [n])."""
        torch = self.torch
        seeds = self.reset_seeds(r)
        act, pixel, any_on = self.actions(r)
        est = torch.where(any_on, 2300, 1300) + torch.where(seeds != 0, 500, 0)   # expected physics steps of the scene in this launch
        order = torch.argsort(est, descending=True, stable=True).to(torch.int32)
        self.sim.set_order_dev(order.data_ptr())
        self.sim.grasp_attempt_reset_dev(act.data_ptr(), reward_row.data_ptr(), seeds.data_ptr(), check_mode=check_mode, table_height=0.91,
                                         settle_ms=1000.0)
        self._alive = (seeds, act, order)                                        # until the next round's launch is queued behind this one
        return act, pixel

    def actions(self, r):
        """[n, 8] f64 action records (x y z rot skip - - -), the aimed pixel index [n] int32 and whether a box is aimed at [n] bool, from the
        CURRENT state on the device."""
        torch = self.torch
        a = torch.zeros((self.n, 8), dtype=torch.float64, device=self.gid.device)
        a[:, 2] = 0.91
        if self.rule == "aimed":
            q = self.state[:, 8:8 + 7 * self.nobj].view(self.n, self.nobj, 7)[:, :, :3] + self.pos0      # [n, nobj, 3] world
            on = (q[:, :, 0].abs() <= 0.27) & ((q[:, :, 1] + 0.6).abs() <= 0.19) & (q[:, :, 2] >= 0.905) & (q[:, :, 2] <= 1.0)
            j = (self.gid + r) % EP
            order = (self.gid[:, None] + j[:, None] + torch.arange(self.nobj, device=on.device)[None, :]) % self.nobj
            on_o = torch.gather(on, 1, order)
            first = torch.argmax(on_o.to(torch.int8), dim=1)                     # first candidate that is on the plate
            pick = torch.gather(order, 1, first[:, None])[:, 0]
            any_on = on_o.any(dim=1)
            xy = q[torch.arange(self.n, device=on.device), pick, :2]
            centre = torch.tensor([0.0, -0.6], dtype=torch.float64, device=on.device)
            a[:, :2] = torch.where(any_on[:, None], xy, centre)
            a[:, 3] = ((self.gid // EP + r) % 6).double()
        else:
            any_on = torch.zeros(self.n, dtype=torch.bool, device=self.gid.device)   # a uniformly drawn pixel rarely has a box under it
            idx = self.table_pixels[torch.randint(len(self.table_pixels), (self.n,), device=self.gid.device, generator=self.gen)]
            a[:, 0] = self.px0[0] + self.dxdpx * (idx % 200).double()
            a[:, 1] = self.px0[1] + self.dydpy * (idx // 200).double()
        px = ((a[:, 0] - self.px0[0]) / self.dxdpx).round().clamp(0, 199)
        py = ((a[:, 1] - self.px0[1]) / self.dydpy).round().clamp(0, 199)
        return a, (py * 200 + px).to(torch.int32), any_on


def cpu_baseline(model, budget_s=12.0, many=False):
    """The fp64 oracle (a port: the reference's own MuJoCo binary cannot exist here) on the host cores, SAME workload as the timed GPU
    rounds: whole episodes of reset + 1000 ms settle + EP aimed grasp attempts (oracle/ur5_oracle.cpp ur5o_batch mode 2, bench_aim),
    one scene per native thread on every core (SURVEY.md section 8d ii), plus the single-core rate."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    if many:
        s1, a1, t1, _, _ = O.batch(model, 1, 0.3 * budget_s, 1, 100)
        sn, an, tn, _, _ = O.batch(model, cores, 0.7 * budget_s, 1, 100)
        return dict(value=sn / tn, unit="env-steps/s", cores=cores, kind="port",
                    sample=f"the first 100 steps of the 40-object drop of scenes 0..{an - 1} ({sn} physics steps), oracle/ur5_oracle.cpp, "
                           f"one scene per thread on {cores} threads for {tn:.1f} s wall",
                    single_core={"value": s1 / t1, "sample": f"{a1} scenes ({s1} steps), {t1:.1f} s on one core"})
    s1, e1, t1, att1, _ = O.batch(model, 1, 0.3 * budget_s, 2, EP)
    sn, en, tn, attn, sucn = O.batch(model, cores, 0.7 * budget_s, 2, EP)
    return dict(value=sn / tn, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{en} whole episodes (scenes 0..{en - 1}: reset + 1000 ms settle + {EP} aimed attempts, the timed GPU rule; {sn} physics "
                       f"steps, {attn} attempts), oracle/ur5_oracle.cpp, one scene per thread on {cores} threads for {tn:.1f} s wall",
                grasp_attempts_per_s=attn / tn, grasp_success_rate=sucn / max(1, attn), env_steps_per_attempt=sn / max(1, attn),
                single_core={"value": s1 / t1, "grasp_attempts_per_s": att1 / t1,
                             "sample": f"{e1} episodes ({s1} steps, {att1} attempts), {t1:.1f} s on one core"})


def rendered_sub_result(torch, dev, local_rank, workload, n, rounds, warmup):
    """Short N=1 measurement of the render + grasp-round shape of BASELINE.json configs[2] (it4) / configs[3] (many) so that the driver's
    one line also times them: `rounds` timed rounds after `warmup`. Every round renders the 200x200 RGB-D observation and aims at the pixel
    over one of the objects in the pick bin (a different one per round); the grasp height comes from the rendered depth at that pixel
    (GraspEnv.step, GraspingEnv.py:100-104), on the device."""
    from mujoco_rl_ur5_amd.model import load_model
    from mujoco_rl_ur5_amd.native import BatchSim
    many = workload == "many"
    model = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml" if many else "/UR5+gripper/UR5gripper_2_finger.xml")
    sim = BatchSim(model, n, device_id=local_rank)
    sim.reset(BASE_SEED + np.arange(n, dtype=np.uint64), 1, 1000.0)
    nobj = (model.nv - 8) // 6
    xpos = sim.body_xpos()[:, 8:8 + nobj]
    from mujoco_rl_ur5_amd.controller import MJ_Controller
    ctl = MJ_Controller(model, sim, None)
    acts = np.zeros((warmup + rounds, n, 8))
    pix = np.zeros((warmup + rounds, n, 2), dtype=np.int64)
    for r in range(warmup + rounds):
        for e in range(n):
            objs = xpos[e]
            inbin = np.where((np.abs(objs[:, 0]) < 0.2) & (np.abs(objs[:, 1] + 0.6) < 0.13) & (objs[:, 2] > 0.85))[0]
            k = inbin[(e + r) % len(inbin)] if len(inbin) else 0
            px, py = ctl.world_2_pixel(objs[k], 200, 200)                      # the pixel over the object's centre
            pix[r, e] = [min(max(int(px), 0), 199), min(max(int(py), 0), 199)]
            acts[r, e, :2] = objs[k, :2]
            acts[r, e, 3] = (e + r) % 6
    actions = torch.from_numpy(acts).to(dev)
    pix_t = torch.from_numpy(pix).to(dev)
    cam_z = float(model.cam_pos0[model.camera_name2id("top_down")][2])
    scene = torch.arange(n, device=dev)
    img = torch.zeros((n, 200, 200, 3), dtype=torch.uint8, device=dev)
    dep = torch.zeros((n, 200, 200), dtype=torch.float32, device=dev)
    reward = torch.zeros((warmup + rounds, n), dtype=torch.int32, device=dev)
    cam = model.camera_name2id("top_down")
    sim.set_stream(torch.cuda.current_stream().cuda_stream)

    def one(r):
        sim.render_dev(img.data_ptr(), dep.data_ptr(), cam, 200, 200, 0)            # get_observation (GraspingEnv.py:390-406), metric depth
        actions[r, :, 2] = cam_z - dep[scene, pix_t[r, :, 1], pix_t[r, :, 0]].double()   # top-down camera: world z of what the pixel shows
        sim.grasp_attempt_dev(actions[r].data_ptr(), reward[r].data_ptr(), check_mode=0, table_height=0.91)
    for r in range(warmup):
        one(r)
    sim.sync()
    c0, k0 = sim.counters(), sim.kernel_ms_total()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(warmup, warmup + rounds):
        one(r)
    sim.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1, k1 = sim.counters(), sim.kernel_ms_total()
    steps = int((c1["total_steps"] - c0["total_steps"]).sum())
    words = model.nq + 2 * model.nv + 5 * model.nu + 8
    out = {"workload": "configs[3] shape: 40-object piles (UR5gripper_2_finger_many_objects.xml, condim 6), 200x200 RGB-D render + grasp script per round" if many
           else "configs[2] shape: IT4 (in-tree UR5gripper_2_finger.xml, 3 boxes + 3 spheres), 200x200 RGB-D render + grasp script per round",
           "scenes": n, "rounds": rounds, "warmup": warmup, "env_steps_per_s": steps / dt, "grasp_attempts_per_s": rounds * n / dt,
           "grasp_success_rate": float(reward[warmup:].sum().item()) / (rounds * n), "env_steps_per_attempt": steps / (rounds * n),
           "newton_iters_per_step": float((c1["solver_iters"] - c0["solver_iters"]).sum()) / max(1, steps),
           "status_bits": int(np.bitwise_or.reduce(c1["status"])), "kernel_ms_per_round": (k1 - k0) / rounds,
           "roofline_frac": steps * 2 * words * 8 / ((k1 - k0) * 1e-3) / 8e12,
           "kernel": "ur5m_run_kernel<248>" if many else "ur5_run_kernel<44>"}
    sim.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--envs", type=int, default=None, help="scenes per GPU (default 4096 for --scaling weak, 4096 / gpus for strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 4096 scenes per GPU; strong: 4096 scenes in total, 4096 / N per GPU (SURVEY.md section 8e)")
    ap.add_argument("--rule", choices=("aimed", "uniform"), default="aimed", help="action rule of the timed rounds (see the module docstring)")
    ap.add_argument("--groups", type=int, default=2, help="scene groups per GPU, one engine handle + HIP stream each, rounds pipelined (1 = one handle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the uniform-rule figure and the it4 / many sub-results (N = 1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for 2 ranks on one device)")
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
    ndev = torch.cuda.device_count()
    dev_id = local_rank % max(1, ndev)                                            # >1 rank per device only in the single-GPU shard-invariance check
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if world > 1:
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))

    model = load_model("it1_4box")
    n_local = args.envs if args.envs else (4096 if args.scaling == "weak" else 4096 // world)
    n_total = n_local * world
    lo, hi = sharding.shard_range(n_total, rank, world)
    G = args.groups if n_local % max(1, args.groups) == 0 else 1
    n_g = n_local // G
    rounds = args.warmup + args.steps

    class Group:
        """One scene group of the rank: engine handle, its HIP stream (torch's, so that action tensors are stream-ordered with the launches)."""
        def __init__(self, g, rule):
            self.lo = lo + g * n_g
            self.sim = BatchSim(model, n_g, device_id=dev_id)
            self.sim.reset(BASE_SEED + np.arange(self.lo, self.lo + n_g, dtype=np.uint64), 1, 1000.0)   # episode 0 (GraspingEnv.py:409-477), untimed
            self.stream = torch.cuda.Stream(device=dev) if G > 1 else torch.cuda.current_stream()
            self.sim.set_stream(self.stream.cuda_stream)
            with torch.cuda.stream(self.stream):
                self.wl = It1Rounds(torch, model, self.sim, dev, self.lo, n_g, n_total, rule)
                self.reward = torch.zeros((rounds + 8, n_g), dtype=torch.int32, device=dev)
                self.ids = torch.arange(self.lo, self.lo + n_g, dtype=torch.int32, device=dev)

    groups = [Group(g, args.rule) for g in range(G)]
    torch.cuda.synchronize()

    def run_rounds(r0, r1, wls=None):
        gathered = None
        for r in range(r0, r1):
            for k, gr in enumerate(groups):
                with torch.cuda.stream(gr.stream):
                    wl = gr.wl if wls is None else wls[k]
                    act, pixel = wl.launch(r, gr.reward[r])                        # attempt (+ reset_model for the scenes whose episode ends)
                    rec = torch.stack([gr.ids, pixel, act[:, 3].to(torch.int32), gr.reward[r]], dim=1)
                    gathered = sharding.gather_outcomes(rec)                       # the path's only collective: 16 B per scene per round
        return gathered

    def counters():
        cs = [gr.sim.counters() for gr in groups]
        return {k: np.concatenate([c[k] for c in cs]) for k in cs[0]}

    def timed(r0, r1, wls=None):
        for gr in groups:
            gr.sim.sync()
        c0, k0 = counters(), sum(gr.sim.kernel_ms_total() for gr in groups)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = run_rounds(r0, r1, wls)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        for gr in groups:
            gr.sim.sync()                                                          # resolves the HIP event pairs of the region's launches
        c1, k1 = counters(), sum(gr.sim.kernel_ms_total() for gr in groups)
        return elapsed, c0, c1, k1 - k0, g

    run_rounds(0, args.warmup)
    elapsed, c0, c1, kernel_ms, gathered = timed(args.warmup, rounds)
    assert gathered.shape == (n_g * world, 4)
    reward = torch.cat([gr.reward[:rounds] for gr in groups], dim=1)
    steps_local = int((c1["total_steps"] - c0["total_steps"]).sum())
    per_round = reward[args.warmup:].double().mean(dim=1)
    elapsed_local = elapsed
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
        # G launches (one per scene group) are in flight at once and share the chip: the rate is taken over the wall time of the timed region,
        # which the engine kernels of the G streams cover back to back (their HIP-event durations are reported next to it)
        achieved = steps_local * bytes_per_step / elapsed_local / 1e9
        traffic, traffic_src = None, "not measured in this run (PMC passes are separate rocprofv3 runs)"
        tp = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
        if os.path.exists(tp):
            with open(tp) as f:
                tj = json.load(f)
            traffic = tj["hbm_bytes_per_env_step"] * steps_local / (args.steps * G)          # per launch, like algorithmic_bytes_per_launch below
            traffic_src = (f"from profiles/ ({tj.get('source', 'hbm_traffic_latest.json')}), NOT measured in this run: (2 x FETCH_SIZE + WRITE_SIZE) per "
                           "env-step of the rocprofv3 PMC passes of this command x env-steps of an average timed launch")
        out = {
            "metric": f"env-steps/sec (+ grasp-attempts/sec), {n_total} parallel UR5 scenes on {world} MI355X",
            "value": steps_all / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "grasp_attempts_per_s": attempts / elapsed, "grasp_success_rate": succ_all / attempts,
            "grasp_success_rate_per_round_rank0": [round(float(x), 4) for x in per_round.tolist()],
            "env_steps_per_attempt": steps_all / attempts,
            "newton_iters_per_step": float((c1["solver_iters"] - c0["solver_iters"]).sum()) / max(1, steps_local),
            "status_bits": int(np.bitwise_or.reduce(c1["status"])),
            "config": {"workload": "BASELINE.json configs[1]: IT1 (UR5gripper_2_finger.xml robot + bins, 4 equal 4 cm boxes), physics only, fixed "
                                   "z = 0.91, lift + 500-step closing check; one grasp-attempt round per step, episodes of 4 rounds with reset_model "
                                   "(+ 1000 ms settle) for the quarter of the batch that starts an episode in the round",
                       "rule": ("aimed: current position of a box still on the pick plate, read from the state records on the device" if args.rule == "aimed"
                                else "uniform: uniformly drawn table pixel, rotation 0 (SURVEY.md 8d / Grasping_Agent_multidiscrete.py:266-280)"),
                       "scenes_per_gpu": n_local, "scenes_total": n_total, "scene_groups_per_gpu": G, "solver": "Newton (MuJoCo default; north_star says PGS, see DESIGN.md D1), "
                       f"tolerance 1e-10, iteration cap {model.opt['iterations']}", "timestep_s": model.opt["timestep"],
                       "parallelism": f"scenes sharded x{world}, 1 all_gather of 16 B outcome records per scene group and round"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "traffic_source": traffic_src, "kernel": "ur5_run_kernel<32>", "bytes_per_env_step": bytes_per_step,
                         "avg_launch_ms": kernel_ms / (args.steps * G), "launches_per_round": G, "launch_concurrency": G,
                         "env_steps_per_launch": steps_local / (args.steps * G),
                         "algorithmic_bytes_per_launch": bytes_per_step * steps_local / (args.steps * G),
                         "kernel_ms_per_round": 1e3 * elapsed_local / args.steps, "env_steps_per_round": steps_local / args.steps,
                         "note": "algorithmic state bytes x env-steps of the timed rounds / their wall time on rank 0: one attempt + episode-reset "
                                 "launch per scene group and round, the groups' launches overlapping on their own HIP streams (avg_launch_ms = "
                                 "HIP-event duration of one launch; achieved = launch_concurrency x bytes per launch / avg_launch_ms up to the "
                                 "overlap at the region's ends). A scene stays in LDS for a whole launch, so the algorithmic figure is an accounting "
                                 "unit, not the traffic: the step is latency / VALU-issue bound (DESIGN.md section 3)"},
        }
        if world == 1 and not args.no_extras:
            # second figure: SURVEY.md 8d's own action rule, same episode structure, a few rounds continuing from the current state
            wls_u = []
            for gr in groups:
                with torch.cuda.stream(gr.stream):
                    wls_u.append(It1Rounds(torch, model, gr.sim, dev, gr.lo, n_g, n_total, "uniform"))
            ur = max(2, min(4, args.steps))                                        # rounds + 1 + ur <= rounds + 8 rows of the reward buffers
            run_rounds(rounds, rounds + 1, wls_u)
            e_u, cu0, cu1, k_u, _ = timed(rounds + 1, rounds + 1 + ur, wls_u)
            st_u = int((cu1["total_steps"] - cu0["total_steps"]).sum())
            succ_u = sum(float(gr.reward[rounds + 1:rounds + 1 + ur].sum().item()) for gr in groups)
            out["uniform_rule"] = {"rounds": ur, "env_steps_per_s": st_u / e_u, "grasp_attempts_per_s": ur * n_local / e_u,
                                   "grasp_success_rate": succ_u / (ur * n_local),
                                   "env_steps_per_attempt": st_u / (ur * n_local),
                                   "rule": "uniformly drawn table pixel, rotation 0, z = 0.91 (SURVEY.md section 8d config 2)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model)
    for gr in groups:
        gr.sim.close()
    if rank == 0 and world == 1 and not args.no_extras:
        torch.cuda.synchronize()
        # the headline line must not depend on the secondary measurements: a failure there is reported in place of the sub-result
        for key, spec in (("it4", ("it4", 4096, 2, 1)), ("many", ("many", 2048, 1, 1))):   # many: BASELINE configs[3], 2048 piles per GPU
            try:
                out[key] = rendered_sub_result(torch, dev, dev_id, *spec)
            except Exception as exc:  # noqa: BLE001
                out[key] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
