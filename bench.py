#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json config 2): 4096 batched IT1 scenes, physics only, fixed-z grasp attempts with the 500-step
closing check (README.md:20 of the reference). A "step" is one grasp-attempt round over the whole batch = the hot path
(GraspEnv.step -> move_and_grasp, GraspingEnv.py:62-156,205-386, plus GraspEnv.reset_model at episode boundaries, :409-477).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...          (no launcher: bench.py starts the N ranks itself through torch.distributed.run, RCCL, 127.0.0.1)

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
For the same reason the rank's scenes are simulated as --groups G scene groups (one engine handle + one HIP stream each, 1 / G of the
scenes each; G = 4 since round 6, one per hardware queue): the next launch of one group is queued behind its current one while the other
groups' launches are still running, so the wave slots that the tail of a launch leaves empty are taken by another group's launch.
Scenes, seeds and results are those of one group.

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
    """Device-side driver of the stationary workloads for one rank: flags / seeds of the scenes that reset, action records.
    kind "it1" (headline, BASELINE configs[1]): physics only, fixed z = 0.91, IT1 closing check (check_mode 1).
    kind "it4" (configs[2]): every round renders the 200x200 RGB-D observation; the grasp height is the rendered depth under the aimed pixel
    (GraspEnv.step, GraspingEnv.py:100-104); in-tree script (check_mode 0). kind "many" (configs[3]): the same on 40-object piles, aimed by the
    box rule of tools/pile_aim.py evaluated on the device (the box with the most level top face whose sides are parallel to the fingers)."""

    def __init__(self, torch, model, sim, dev, lo, n_local, n_total, rule, kind="it1"):
        from mujoco_rl_ur5_amd.controller import MJ_Controller
        self.torch, self.sim, self.rule_name, self.n, self.n_total, self.kind = torch, sim, rule, n_local, n_total, kind
        self.gid = torch.arange(lo, lo + n_local, dtype=torch.int64, device=dev)
        self.gid0 = int(lo)   # (host copy: reading gid[0] back per launch is a blocking device-to-host copy BETWEEN two engine launches of the stream -- profiles/r06_x_headline_trace_finding.txt)
        self.state = sim.state_tensor(dev)                                        # [n, stride] f64, aliases the engine's records
        # objects: 3 slides + ball each (UR5gripper_2_finger.xml:233-239): world position = body_pos + slide offsets; free joints hold world coordinates
        self.nobj = (model.nq - 8) // 7
        obj_bodies = [int(model.jnt_bodyid[list(model.jnt_qposadr).index(8 + 7 * k)]) for k in range(self.nobj)]
        free = int(model.jnt_type[list(model.jnt_qposadr).index(8)]) == 0
        self.pos0 = torch.from_numpy(np.zeros((self.nobj, 3)) if free else np.asarray(model.body_pos)[obj_bodies].copy()).to(dev)   # [nobj, 3]
        self.cam_id = model.camera_name2id("top_down")
        self.cam_z = float(model.cam_pos0[self.cam_id][2])
        if kind != "it1":
            self.img = torch.zeros((n_local, 200, 200, 3), dtype=torch.uint8, device=dev)
            self.dep = torch.zeros((n_local, 200, 200), dtype=torch.float32, device=dev)
        if kind == "many":
            g0 = model.ngeom - self.nobj
            self.box_idx = torch.tensor([k for k in range(self.nobj) if int(model.geom_type[g0 + k]) == 6], dtype=torch.int64, device=dev)
            self.box_half = torch.from_numpy(np.asarray(model.geom_size)[g0:g0 + self.nobj][self.box_idx.cpu().numpy()].copy()).to(dev)   # [nb, 3]
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

    def launch(self, r, reward_row, check_mode=None):
        """One round on the engine's stream: (observation,) action records from the current state, longest-expected-first dispatch order, ONE
        launch of attempt (+ episode reset). Returns (action records [n, 8], aimed pixel [n])."""
        torch = self.torch
        if check_mode is None:
            check_mode = 1 if self.kind == "it1" else 0
        seeds = self.reset_seeds(r)
        if self.kind != "it1":
            self.sim.render_dev(self.img.data_ptr(), self.dep.data_ptr(), self.cam_id, 200, 200, 0)   # get_observation (GraspingEnv.py:390-406), metric depth
        act, pixel, any_on = self.actions(r)
        if self.kind != "it1":                                                   # top-down camera: world z of what the aimed pixel shows (GraspingEnv.py:100-104)
            px = pixel.long()
            act[:, 2] = self.cam_z - self.dep[torch.arange(self.n, device=px.device), px // 200, px % 200].double()
        est = torch.where(any_on, 2300, 1300) + torch.where(seeds != 0, 500, 0)   # expected physics steps of the scene in this launch
        if os.environ.get("UR5_BENCH_ORDER_RESETS_ONLY"):                          # experiment: what the order is worth (the in-kernel rule cannot know `any_on` before the launch)
            est = torch.where(seeds != 0, 500, 0)
        order = torch.argsort(est, descending=True, stable=True).to(torch.int32)
        self.sim.set_order_dev(order.data_ptr())
        self.sim.grasp_attempt_reset_dev(act.data_ptr(), reward_row.data_ptr(), seeds.data_ptr(), check_mode=check_mode, table_height=0.91,
                                         settle_ms=1000.0)
        self._alive = (seeds, act, order)                                        # until the next round's launch is queued behind this one
        return act, pixel

    def rule(self):
        """The aiming rule as the engine evaluates it in the kernel (include/ur5sim.h ur5_aim_rule): exactly `actions()` below for rule "aimed" -- kind 1 (an object still on
        the pick plate) for the small scenes, kind 2 (the pile box rule) for 40-object piles; the rendered workloads take the grasp height from the depth image the scene
        renders for itself at the start of the round (z_from_depth, ur5_set_observation_dev), through the top-down camera's pixel map at table height."""
        from mujoco_rl_ur5_amd.native import AimRule
        cam = {} if self.kind == "it1" else dict(z_from_depth=1, cam_x0=self.px0[0], cam_y0=self.px0[1], cam_dx=self.dxdpx, cam_dy=self.dydpy, cam_z=self.cam_z)
        return AimRule(kind=2 if self.kind == "many" else 1, episode_rounds=EP, first_scene_id=self.gid0, n_total=int(self.n_total), base_seed=BASE_SEED, plate_half_x=0.27,
                       plate_centre_y=-0.6, plate_half_y=0.19, z_min=0.905, z_max=1.0, grasp_z=0.91, fallback_x=0.0, fallback_y=-0.6, **cam)

    def plan_rounds(self, r0, r1, fused):
        """Everything the launches of rounds r0 .. r1 - 1 need besides the engine, for ALL of them at once: the action-record buffer the kernel fills and one dispatch order
        per launch (the scenes with more episode ends inside the launch first), a handful of torch kernels queued BEFORE the first engine launch. Round 5 did this per
        launch: the zero-fill of one launch's records then sat between two engine launches of the stream and waited for a free wave slot while the OTHER group's launch
        held every register of the chip -- 0.8 s for a 512 KB fill in the rocprofv3 trace (profiles/r05_x_kernel_stats.csv), the group's next launch behind it."""
        torch = self.torch
        assert self.rule_name == "aimed", "the rules the kernel evaluates are the aimed ones"
        starts = list(range(r0, r1, fused))
        act = torch.zeros((r1 - r0, self.n, 8), dtype=torch.float64, device=self.gid.device)
        rr = torch.arange(r0, r1, device=self.gid.device)
        ends = (((self.gid[:, None] + rr[None, :] + 1) % EP) == 0).to(torch.int32)                   # [n, rounds]: the scene's episode ends after that round
        stops = starts[1:] + [r1]
        resets = torch.stack([ends[:, s - r0:e - r0].sum(dim=1) for s, e in zip(starts, stops)])      # [launches, n]
        order = torch.argsort(resets, dim=1, descending=True, stable=True).to(torch.int32).contiguous()
        plan = dict(r0=r0, r1=r1, starts=starts, act=act, order=order)
        if self.kind != "it1":
            # the rendered workloads (round 6): every scene renders its own 200x200 RGB-D observation at the start of each of its rounds, inside the launch
            # (ur5_set_observation_dev); one frame per round of a launch, so that a launch leaves the observations of all its rounds behind
            frames = min(fused, r1 - r0)
            if getattr(self, "_frames", None) is None or self._frames[0].shape[0] < frames:
                self._frames = (torch.zeros((frames, self.n, 200, 200, 3), dtype=torch.uint8, device=self.gid.device),
                                torch.zeros((frames, self.n, 200, 200), dtype=torch.float32, device=self.gid.device))
            self.sim.set_observation_dev(self._frames[0].data_ptr(), self._frames[1].data_ptr(), self.cam_id, 200, 200, frames=int(self._frames[0].shape[0]))
            plan["frames"] = self._frames
        return plan

    def launch_planned(self, plan, i, reward):
        """Launch i of a plan: rounds starts[i] .. of every scene in ONE launch, no lock step between scenes (ur5_grasp_rounds_dev): the scene aims by itself with rule(),
        attempts, and resets + settles where its episode ends, then goes on to its next round. Per-scene results are bit-identical to lock-step calls of launch()
        (tests/test_grasp_rounds.py). Nothing but the engine kernel is queued: the order is read in place (ur5_set_order_view_dev), the records were zeroed by the plan."""
        s0 = plan["starts"][i]
        k = min(plan["starts"][i + 1] if i + 1 < len(plan["starts"]) else plan["r1"], plan["r1"]) - s0
        self.sim.set_order_view_dev(plan["order"][i].data_ptr())
        self.sim.grasp_rounds_dev(self.rule(), s0, k, reward[s0:s0 + k].data_ptr(), plan["act"][s0 - plan["r0"]:s0 - plan["r0"] + k].data_ptr(),
                                  check_mode=1 if self.kind == "it1" else 0, table_height=0.91, settle_ms=1000.0)

    def planned_pixels(self, plan):
        """aimed pixel [rounds, n] int32 of the action records the kernel wrote"""
        act = plan["act"]
        px = ((act[..., 0] - self.px0[0]) / self.dxdpx).round().clamp(0, 199)
        py = ((act[..., 1] - self.px0[1]) / self.dydpy).round().clamp(0, 199)
        return (py * 200 + px).to(self.torch.int32)

    def launch_rounds(self, r0, k, reward_rows):
        """Rounds r0 .. r0 + k - 1 in ONE launch (a plan of one launch). reward_rows: int32 [k, n] (contiguous rows of the reward buffer).
        Returns (action records [k, n, 8], aimed pixels [k, n] int32)."""
        plan = self.plan_rounds(r0, r0 + k, k)
        self.sim.set_order_view_dev(plan["order"][0].data_ptr())
        self.sim.grasp_rounds_dev(self.rule(), r0, k, reward_rows.data_ptr(), plan["act"].data_ptr(), check_mode=1 if self.kind == "it1" else 0, table_height=0.91, settle_ms=1000.0)
        self._alive = (plan, reward_rows)
        return plan["act"], self.planned_pixels(plan)

    def actions(self, r):
        """[n, 8] f64 action records (x y z rot skip - - -), the aimed pixel index [n] int32 and whether a box is aimed at [n] bool, from the
        CURRENT state on the device."""
        torch = self.torch
        a = torch.zeros((self.n, 8), dtype=torch.float64, device=self.gid.device)
        a[:, 2] = 0.91
        if self.rule_name == "aimed" and self.kind == "many":
            xy, rot, any_on = self.pile_box_actions(r)
            a[:, :2] = xy
            a[:, 3] = rot.double()
        elif self.rule_name == "aimed":
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


    def pile_box_actions(self, r):
        """tools/pile_aim.py pick_box for every scene at once, on the device: (xy [n, 2], rotation index [n], a box was found [n]). Scenes without a box
        in the bin aim at their highest object in the bin (rotation cycling), an empty bin gets an attempt at its centre."""
        torch = self.torch
        n, dev = self.n, self.gid.device
        P = self.state[:, 8:8 + 7 * self.nobj].view(n, self.nobj, 7)
        c = P[:, :, :3]                                                          # [n, nobj, 3] world (free joints)
        q = P[:, self.box_idx, 3:7]
        w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).view(n, -1, 3, 3)
        cb = c[:, self.box_idx]                                                  # [n, nb, 3]
        cosv, a_ax = R[:, :, 2, :].abs().max(dim=-1)                             # the box axis nearest to the vertical
        tilt = torch.rad2deg(torch.acos(cosv.clamp(max=1.0)))
        b_ax = ((a_ax + 1) % 3)[..., None, None].expand(-1, -1, 3, 1)
        bcol = torch.gather(R, 3, b_ax)[..., 0]                                  # world direction of the next axis: [n, nb, 3]
        want = -torch.rad2deg(torch.atan2(bcol[..., 1], bcol[..., 0]))           # fingers parallel to the box's sides: wrist angle = -yaw (mod 90)
        ang = torch.tensor([0.0, 30.0, -30.0], dtype=torch.float64, device=dev)  # rotation indices 0 / 1 / 4 (GraspingEnv.py:40)
        mis = (torch.remainder(want[..., None] - ang + 45.0, 90.0) - 45.0).abs() # [n, nb, 3]
        mis_min, which = mis.min(dim=-1)
        d = c[:, None, :, :] - cb[:, :, None, :]                                 # [n, nb, nobj, 3]
        other = torch.arange(self.nobj, device=dev)[None, None, :] != self.box_idx[None, :, None]
        on_top = ((torch.hypot(d[..., 0], d[..., 1]) < 0.05) & (d[..., 2] > 0.01) & other).any(dim=-1)
        inbin = (cb[..., 0].abs() < 0.17) & ((cb[..., 1] + 0.6).abs() < 0.10) & (cb[..., 2] > 0.89)
        score = tilt + mis_min + 100.0 * on_top.double() + 1e6 * (~inbin).double()
        best_score, best = score.min(dim=1)
        e = torch.arange(n, device=dev)
        found = best_score < 1e5
        rot_box = torch.tensor([0, 1, 4], dtype=torch.int64, device=dev)[which[e, best]]
        # fallback: the highest object inside the bin
        ib = (c[..., 0].abs() < 0.2) & ((c[..., 1] + 0.6).abs() < 0.13) & (c[..., 2] > 0.85)
        top = torch.where(ib, c[..., 2], torch.full_like(c[..., 2], -1.0)).argmax(dim=1)
        any_obj = ib.any(dim=1)
        centre = torch.tensor([0.0, -0.6], dtype=torch.float64, device=dev)
        xy_fb = torch.where(any_obj[:, None], c[e, top, :2], centre)
        xy = torch.where(found[:, None], cb[e, best, :2], xy_fb)
        rot = torch.where(found, rot_box, (self.gid // EP + r) % 6)
        return xy, rot, found | any_obj


def cpu_baseline(model, budget_s=12.0, kind="it1"):
    """The fp64 oracle (a port: the reference's own MuJoCo binary cannot exist here) on the host cores, SAME workload as the timed GPU
    rounds: whole episodes of reset + 1000 ms settle + EP aimed grasp attempts (oracle/ur5_oracle.cpp ur5o_batch mode 2 / 3, bench_aim; kind "it4"
    renders the observation and takes the grasp height from its depth image, like the GPU rounds), one scene per native thread on every core
    (SURVEY.md section 8d ii), plus the single-core rate. kind "many": one thread per scene does reset + settle + ONE rendered attempt on a
    40-object pile (a whole episode of a pile is minutes of CPU time; the sample is bounded, its physics steps are the same kind of steps)."""
    from oracle import oracle as O
    cores = usable_cores()
    if kind == "many":
        sn, an, tn, attn, sucn = O.batch(model, cores, budget_s, 4, 1)              # scenes in flight when the budget ends are cut off; their steps count
        return dict(value=sn / tn, unit="env-steps/s", cores=cores, kind="port",
                    sample=f"{an} piles (scenes 0..{an - 1}): reset + 1000 ms settle + one rendered grasp attempt each, aimed by the timed GPU rounds' box rule (oracle bench_pile_aim == It1Rounds.pile_box_actions), cut off "
                           f"after {budget_s:.0f} s ({sn} physics steps, {attn} attempts completed), oracle/ur5_oracle.cpp, one scene per thread on {cores} threads, {tn:.1f} s wall",
                    grasp_attempts_per_s=attn / tn, grasp_success_rate=sucn / max(1, attn), env_steps_per_attempt=sn / max(1, attn))
    mode = 2 if kind == "it1" else 3
    s1, e1, t1, att1, _ = O.batch(model, 1, 0.3 * budget_s, mode, EP)
    sn, en, tn, attn, sucn = O.batch(model, cores, 0.7 * budget_s, mode, EP)
    what = "aimed attempts, the timed GPU rule" if kind == "it1" else "aimed attempts with a rendered 200x200 RGB-D observation and depth-derived grasp height each, the timed GPU rule"
    return dict(value=sn / tn, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{en} whole episodes (scenes 0..{en - 1}: reset + 1000 ms settle + {EP} {what}; {sn} physics "
                       f"steps, {attn} attempts), oracle/ur5_oracle.cpp, one scene per thread on {cores} threads for {tn:.1f} s wall",
                grasp_attempts_per_s=attn / tn, grasp_success_rate=sucn / max(1, attn), env_steps_per_attempt=sn / max(1, attn),
                single_core={"value": s1 / t1, "grasp_attempts_per_s": att1 / t1,
                             "sample": f"{e1} episodes ({s1} steps, {att1} attempts), {t1:.1f} s on one core"})


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota). A GPU box
    reports 256 hardware threads but may grant the container far fewer; oversubscribing the oracle's threads would only stretch the sample."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(math.ceil(int(q) / int(per)))))
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(math.ceil(q / per))))
        except (OSError, ValueError):
            pass
    return n


class Job:
    """The scenes of one rank as G scene groups (one engine handle + one HIP stream each, rounds pipelined across the groups) and the timed-region
    protocol of the driver contract: barrier + synchronize on both sides, exactly the rounds asked for in between."""

    def __init__(self, torch, dist, sharding, model, kind, rule, n_local, n_total, lo, world, dev, dev_id, G, rows):
        from mujoco_rl_ur5_amd.native import BatchSim
        from mujoco_rl_ur5_amd.streams import group_streams
        self.torch, self.dist, self.sharding, self.world, self.model, self.kind = torch, dist, sharding, world, model, kind
        self.G = G if n_local % max(1, G) == 0 else 1
        self.n_g, self.n_local, self.n_total = n_local // self.G, n_local, n_total
        job = self

        class Group:
            """One scene group: engine handle, its HIP stream (torch's, so that action tensors are stream-ordered with the launches)."""
            def __init__(self, g):
                self.lo = lo + g * job.n_g
                self.sim = BatchSim(model, job.n_g, device_id=dev_id)
                self.sim.reset(BASE_SEED + np.arange(self.lo, self.lo + job.n_g, dtype=np.uint64), 1, 1000.0)   # episode 0 (GraspingEnv.py:409-477), untimed
                self.stream = job.streams[g]
                self.sim.set_stream(self.stream.cuda_stream)
                with torch.cuda.stream(self.stream):
                    self.wl = It1Rounds(torch, model, self.sim, dev, self.lo, job.n_g, n_total, rule, kind)
                    self.reward = torch.zeros((rows, job.n_g), dtype=torch.int32, device=dev)
                    self.ids = torch.arange(self.lo, self.lo + job.n_g, dtype=torch.int32, device=dev)
        # streams that the runtime has put on different hardware queues -- measured, not assumed (mujoco_rl_ur5_amd/streams.py)
        self.streams, self.streams_overlap_verified = group_streams(torch, dev, self.G, first_high_priority=bool(int(os.environ.get("UR5_GROUP0_HIGH_PRIORITY", "0"))))
        self.groups = [Group(g) for g in range(self.G)]
        torch.cuda.synchronize()

    def run_rounds(self, r0, r1, wls=None, fused=0):
        """Rounds r0 .. r1 - 1. fused = K > 0 (headline workload only): K consecutive rounds per launch with the aiming rule evaluated in the kernel -- a scene does not
        wait for the others between its rounds (ur5_grasp_rounds_dev), and the outcome records of the K rounds travel in ONE all_gather per launch."""
        torch, gathered = self.torch, None
        self.launches = getattr(self, "launches", 0)
        if fused > 0 and wls is None:
            # (1) every group's plan (zeroed records, dispatch orders) while the chip is idle, (2) ALL engine launches of the region, group by group inside a launch index, with
            # nothing between two launches of a stream but an event record, (3) the outcome records of the whole region in one pass per group and ONE all_gather per group
            # (16 B per scene and round). With K rounds per launch an outcome becomes visible to the other ranks when its region ends, not its launch: the scripted-policy
            # rollout (example_agent.py:15-27) reads no outcome; the learner loop (agent.py) gathers per round.
            plans, self.launch_events = [], [[] for _ in self.groups]
            for g, gr in enumerate(self.groups):
                with torch.cuda.stream(gr.stream):
                    plans.append(gr.wl.plan_rounds(r0, r1, fused))
            for i in range(max(len(p["starts"]) for p in plans)):
                for g, gr in enumerate(self.groups):
                    if i >= len(plans[g]["starts"]):
                        continue
                    with torch.cuda.stream(gr.stream):
                        gr.wl.launch_planned(plans[g], i, gr.reward)
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(gr.stream)
                        self.launch_events[g].append(ev)
                        self.launches += 1
            for g, gr in enumerate(self.groups):
                with torch.cuda.stream(gr.stream):
                    k = r1 - r0
                    pixel = gr.wl.planned_pixels(plans[g])
                    rec = torch.stack([gr.ids[None, :].expand(k, -1), pixel, plans[g]["act"][..., 3].to(torch.int32), gr.reward[r0:r1]], dim=2).reshape(-1, 4)
                    gathered = self.sharding.gather_outcomes(rec)                  # [world * rounds * n_g, 4]
            self._plans = plans                                                    # alive until the region's work has run
            return gathered
        for r in range(r0, r1):
            for k, gr in enumerate(self.groups):
                with torch.cuda.stream(gr.stream):
                    wl = gr.wl if wls is None else wls[k]
                    act, pixel = wl.launch(r, gr.reward[r])                        # (observation +) attempt (+ reset_model for the scenes whose episode ends)
                    rec = torch.stack([gr.ids, pixel, act[:, 3].to(torch.int32), gr.reward[r]], dim=1)
                    gathered = self.sharding.gather_outcomes(rec)                  # the path's only collective: 16 B per scene per round
                    self.launches += 1
        return gathered

    def counters(self):
        cs = [gr.sim.counters() for gr in self.groups]
        return {k: np.concatenate([c[k] for c in cs]) for k in cs[0]}

    def timed(self, r0, r1, wls=None, fused=0):
        torch = self.torch
        for gr in self.groups:
            gr.sim.sync()
        c0, k0 = self.counters(), sum(gr.sim.kernel_ms_total() for gr in self.groups)
        if self.sharding.collectives_active():
            self.dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        l0 = getattr(self, "launches", 0)
        g = self.run_rounds(r0, r1, wls, fused)
        self.timed_launches = self.launches - l0
        torch.cuda.synchronize()
        if self.sharding.collectives_active():
            self.dist.barrier()
        elapsed = time.perf_counter() - t0
        for gr in self.groups:
            gr.sim.sync()                                                          # resolves the HIP event pairs of the region's launches
        c1, k1 = self.counters(), sum(gr.sim.kernel_ms_total() for gr in self.groups)
        # What the kernel sustains once both groups' launches overlap, next to the driver-protocol figure (which also pays for the region's two edges: the second group's
        # first launch only starts when the first group's workgroups begin to retire, and the region ends with one group draining alone): per group, the env-steps of its
        # launches 2 .. L over the time between the END of its first and the END of its last launch (HIP events on its stream); the groups run side by side, so the rates add.
        # The workload is stationary by construction (a quarter of the scenes starts an episode in every round), so a launch's share of the region's steps is 1 / L.
        self.steady_state_env_steps_per_s = None
        evs = getattr(self, "launch_events", None)
        if fused > 0 and evs and all(len(e) >= 3 for e in evs):
            rate = 0.0
            for gi, gr in enumerate(self.groups):
                L = len(evs[gi])
                sl = slice(gi * self.n_g, (gi + 1) * self.n_g)
                steps_g = float((c1["total_steps"][sl] - c0["total_steps"][sl]).sum())
                rate += steps_g * (L - 1) / L / (1e-3 * evs[gi][0].elapsed_time(evs[gi][-1]))
            self.steady_state_env_steps_per_s = rate
        self.launch_events = None
        return elapsed, c0, c1, k1 - k0, g

    def close(self):
        for gr in self.groups:
            gr.sim.close()


def rendered_sub_result(torch, dist, sharding, dev, dev_id, workload, n, rounds, warmup, with_cpu, groups=2, fused=1):
    """N = 1 measurement of BASELINE.json configs[2] (it4) / configs[3] (many) in the driver's one line, with the headline's machinery: the same
    stationary episode structure (EP rounds, a quarter of the batch resets per round inside the attempt's launch), every round renders the
    200x200 RGB-D observation, re-aims ON THE DEVICE from the current state, takes the grasp height from the rendered depth under the aimed pixel
    (GraspEnv.step, GraspingEnv.py:100-104), dispatches longest-expected-first, two scene groups pipelined; `rounds` timed rounds after `warmup`."""
    from mujoco_rl_ur5_amd.model import load_model
    many = workload == "many"
    model = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml" if many else "/UR5+gripper/UR5gripper_2_finger.xml")
    job = Job(torch, dist, sharding, model, workload, "aimed", n, n, 0, 1, dev, dev_id, groups, warmup + rounds)
    # fused = K > 0 (round 6): K rounds per launch, every scene renders its own observation and aims inside the launch (ur5_set_observation_dev, rule kind 1 + depth /
    # kind 2): no stand-alone render and no torch kernel between two launches of a stream. 0: the round-5 shape (render, torch rule, one launch per round).
    job.run_rounds(0, warmup, None, fused)
    dt, c0, c1, kms, _ = job.timed(warmup, warmup + rounds, None, fused)
    steps = int((c1["total_steps"] - c0["total_steps"]).sum())
    succ = sum(float(gr.reward[warmup:warmup + rounds].sum().item()) for gr in job.groups)
    words = model.nq + 2 * model.nv + 5 * model.nu + 8
    out = {"workload": ("BASELINE.json configs[3] shape: 40-object piles (UR5gripper_2_finger_many_objects.xml, condim 6), per round a 200x200 RGB-D render, the "
                        "box rule of tools/pile_aim.py on the device, depth-derived grasp height, one grasp script; episodes of 4 rounds with reset_model + 1000 ms settle" if many
                        else "BASELINE.json configs[2] shape: IT4 (in-tree UR5gripper_2_finger.xml, 3 boxes + 3 spheres), per round a 200x200 RGB-D render, device-side "
                        "re-aim at an object still on the plate, depth-derived grasp height, one grasp script; episodes of 4 rounds with reset_model + 1000 ms settle"),
           "scenes": n, "rounds": rounds, "warmup": warmup, "scene_groups": job.G, "rounds_per_launch": fused if fused else 1,
           "observation": ("rendered by every scene for itself at the start of each of its rounds, inside the launch (ur5_set_observation_dev); rule evaluated in the kernel" if fused
                           else "stand-alone render + torch rule between the launches (round-5 shape)"), "scene_group_streams_overlap_verified": job.streams_overlap_verified, "env_steps_per_s": steps / dt, "grasp_attempts_per_s": rounds * n / dt,
           "grasp_success_rate": succ / (rounds * n), "env_steps_per_attempt": steps / (rounds * n),
           "newton_iters_per_step": float((c1["solver_iters"] - c0["solver_iters"]).sum()) / max(1, steps),
           "status_bits": int(np.bitwise_or.reduce(c1["status"] | c1["status_ended"])), "scenes_flagged": int(((c1["status"] | c1["status_ended"]) != 0).sum()),
           "ms_per_round": 1e3 * dt / rounds, "kernel_ms_per_round_and_group": kms / (rounds * job.G),
           "roofline_frac": steps * 2 * words * 8 / dt / 8e12, "bytes_per_env_step": 2 * words * 8,
           "kernel": "ur5m_run_kernel<248>" if many else "ur5_run_kernel<44>"}
    reward_round0 = torch.cat([gr.reward[0] for gr in job.groups]).cpu().numpy()   # the warm-up round = every scene's FIRST attempt after reset_model: what the CPU leg's piles do
    job.close()
    # counter traffic of this kernel, from the round's rocprofv3 PMC passes of `bench.py --sub many` (tools/gpu_evidence_extras.sh): not measured in this run
    tp = os.path.join(ROOT, "profiles", "many_hbm_traffic_latest.json")
    if many and os.path.exists(tp):
        with open(tp) as f:
            tj = json.load(f)
        out["hbm_bytes_per_env_step_from_profiles"] = tj["hbm_bytes_per_env_step"]
        out["traffic_over_algorithmic"] = tj["hbm_bytes_per_env_step"] / (2 * words * 8)
        out["traffic_source"] = f"profiles/many_hbm_traffic_latest.json ({tj.get('source', '')}): 2 x FETCH_SIZE + WRITE_SIZE per env-step, separate PMC passes; NOT measured in this run"
    if with_cpu:
        out["cpu_baseline"] = cb = cpu_baseline(model, 20.0 if many else 8.0, workload)
        if many:
            # the same (scene, round 0) pairs on the GPU (round-5 verdict 1g: the CPU leg completed 12 - 15 attempts with 0 successes two rounds in a row against the GPU
            # rounds' 0.19 -- P ~ 8 % by chance): scenes 0 .. k - 1, first attempt after reset_model + settle, the same box rule on both sides
            k_done, k_started = int(round(cb["grasp_attempts_per_s"] * 20.0)), int(cb["sample"].split(" piles")[0])
            cb["gpu_rewards_same_scenes_round0"] = {"scenes_started_on_the_cpu": k_started, "gpu_successes_among_them": int(reward_round0[:k_started].sum()),
                                                    "gpu_success_rate_round0_all_scenes": float(reward_round0.mean()),
                                                    "note": f"the CPU leg completed about {k_done} of these attempts inside its budget (which ones depends on thread timing); the GPU's "
                                                            "round-0 rewards of the same scene ids are listed so that a CPU leg without a success can be read against them",
                                                    "gpu_rewards_first_scenes": reward_round0[:min(k_started, 64)].astype(int).tolist()}
    return out


def dqn_sub_result(torch, dev, dev_id, n, rounds, warmup, groups=2):
    """BASELINE.json configs[4] shape on one GPU (the 8-GPU version shards the scenes and all-gathers the 16-byte outcome records,
    mujoco_rl_ur5_amd/agent.py): per round render -> transform_observation (+ colour jitter) -> pixel-wise grasp-Q CNN forward for every scene
    (Modules.py MULTIDISCRETE_RESNET, fp32) -> epsilon-greedy action -> grasp script -> replay push + optimiser steps, all device-resident."""
    from mujoco_rl_ur5_amd.agent import BatchedGraspAgent
    # two scene groups per rank (round 5): group 1's render -> CNN forward -> action selection runs under group 0's grasp launch, a group's replay pushes and
    # optimiser steps under the next group's launch; the transitions, batches and optimiser steps are those of the unpipelined loop (tests/test_agent.py)
    from mujoco_rl_ur5_amd import sharding as _sh0
    _sh0.TIME_COLLECTIVES = _sh0.collectives_active()
    agent = BatchedGraspAgent(n_envs=n, device=dev, max_updates_per_round=16, pipeline_groups=groups, device_id=dev_id)
    for e in agent.envs:
        e.reset()
    total = lambda: sum(int(e.sim.counters()["total_steps"].sum()) for e in agent.envs)
    for _ in range(warmup):
        agent.round()
    torch.cuda.synchronize()
    c0 = total()
    u0, t0 = agent.learner.updates_done, time.perf_counter()
    rew = 0.0
    for _ in range(rounds):
        out = agent.round()
        rew += float(out["reward"].float().mean())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = total() - c0
    res = {"workload": "BASELINE.json configs[4] shape on 1 GPU: 40-object piles, 200x200 RGB-D render, grasp-Q CNN (7.32 M parameters, fp32) forward for every scene, "
                       "epsilon-greedy pixel + rotation, grasp script, replay push + optimiser steps (batch 12), device-resident",
           "scenes": n, "rounds": rounds, "warmup": warmup, "pipelined_scene_groups": groups, "grasp_attempts_per_s": rounds * n / dt, "env_steps_per_s": steps / dt, "ms_per_round": 1e3 * dt / rounds,
           "optimiser_steps_per_round": (agent.learner.updates_done - u0) / rounds, "update_to_data": out["update_to_data"], "grasp_success_rate": rew / rounds,
           "epsilon": out["epsilon"], "loss": out["loss"], "cnn_gflop_per_scene_forward": 42.0}
    from mujoco_rl_ur5_amd import sharding as _sh
    if _sh.collectives_active():
        # what the multi-rank agent path adds per round (round-5 verdict 7b): the flattened weight / Adam-state broadcast, the replay batch's all-reduce per optimiser step, the
        # outcome all_gather -- issued here by ONE rank through RCCL (--collectives --backend nccl): device time of RCCL's own kernels; the wire estimate is per xGMI link
        st = _sh.collective_stats()
        tot_rounds = rounds + warmup
        res["collectives_per_round"] = {k: dict(calls=v["calls"] / tot_rounds, mbytes=v["bytes"] / tot_rounds / 1e6, device_ms=v["ms"] / tot_rounds) for k, v in st.items()}
        bw = 153e9                                                                  # one xGMI link, bytes/s (7 per GPU; ring collectives are per-link bound)
        b, a = st.get("broadcast", dict(bytes=0, calls=0)), st.get("all_reduce_replay_batch", dict(bytes=0, calls=0))
        res["collectives_xgmi_estimate_ms_per_round_8_gpus"] = dict(
            broadcast=(b["bytes"] / tot_rounds) / bw * 1e3, all_reduce=2 * 7 / 8 * (a["bytes"] / tot_rounds) / bw * 1e3 + a["calls"] / tot_rounds * 14 * 0.02,
            note="ring estimate on one 153 GB/s link: broadcast = bytes / link rate (pipelined ring); all-reduce = 2 (N-1)/N x bytes / link rate + 2 (N-1) hops x ~20 us per call")
    res["learning_cadence"] = (f"{res['optimiser_steps_per_round']:.0f} optimiser steps per round of {n} transitions (update_to_data {res['update_to_data']:.4f}); the reference takes one step per "
                               "transition (Grasping_Agent_multidiscrete.py:551-556): max_updates_per_round caps the replicated learner's share of a round")
    for e in agent.envs:
        e.sim.close()
    return res


# rounds per launch of the rendered sub-results (the observation rendered and the rule evaluated inside the launch, round 6): same-box sweeps in profiles/r06_*_rendered_rounds_per_launch.log
# Piles: lock-step launches (K = 0) -- 679 / 681 / 679 / 680 k env-steps/s against 652 k for K = 1 on one box (profiles/r06_o_many_rounds_per_launch.log). The two groups' K = 1
# launches start together and stay in phase: both drain at once and every round has a tail nobody fills; the lock-step shape's small kernels between two launches of a stream
# offset the groups by chance (a 1 s spin in front of group 1's first launch gives K = 1 the same 680 k). What the dispatch order is worth there: nothing (resets-only order: 681 / 680 k).
SUB_FUSED = {"it4": 2, "many": 0, "many4096": 0}


def default_rounds_per_launch(n_local):
    """K of ur5_grasp_rounds_dev by the scenes per GPU, from same-box sweeps on the MI355X (round 6, no torch kernel between two launches of a stream any more:
    profiles/r06_g_headline_rounds_per_launch.log). A full chip (2 x 2048 scenes) wants SHORT launches -- each group's tail is refilled by the other group's next launch, and
    the region's last tail is 2 rounds long instead of 4: K = 2 gives 18.4 M env-steps/s, K = 4 17.6 M, K = 8 16.6 M. A shard that cannot fill the chip wants LONG ones (its
    only tail is the launch's end): 2048 scenes K = 8, 512 scenes K = 16 (4.4 M against 4.0 M at K = 8). What K costs: an outcome record is gathered when its launch region
    has run, i.e. up to K rounds after the attempt (the scripted-policy rollout reads no outcome; the learner loop of agent.py gathers every round)."""
    return 2 if n_local >= 4096 else (8 if n_local >= 1024 else 16)


def strong_scaling_points(torch, dist, sharding, model, dev, dev_id, args):
    """What one GPU does with its share of a STRONG-scaling run (4096 scenes in total: 2048 / 1024 / 512 per GPU at 2 / 4 / 8 GPUs), measured here on this
    GPU alone with the headline's rounds (no data-path collective: a shard is independent). Below 2048 scenes the wave slots are no longer full (2048 per
    chip) and a round lasts as long as its slowest scene, so the rate falls with the scene count -- DESIGN.md section 6 has the numbers and the limit."""
    pts = []
    for n in (2048, 1024, 512):
        try:
            k = 0 if args.fused_rounds == 0 else default_rounds_per_launch(n)
            job = Job(torch, dist, sharding, model, "it1", "aimed", n, n, 0, 1, dev, dev_id, 2, 20)   # two groups: the shards' sweeps (profiles/r06_g_*) were taken so
            job.run_rounds(0, 2, None, k)
            dt, c0, c1, _, _ = job.timed(2, 18, None, k)
            steps = int((c1["total_steps"] - c0["total_steps"]).sum())
            pts.append({"scenes_per_gpu": n, "env_steps_per_s_per_gpu": steps / dt, "ms_per_round": 1e3 * dt / 16, "rounds": 16, "rounds_per_launch": k or 1,
                        "corresponds_to": f"{4096 // n} GPUs x {n} scenes"})
            job.close()
        except Exception as exc:  # noqa: BLE001
            pts.append({"scenes_per_gpu": n, "error": f"{type(exc).__name__}: {exc}"})
    return pts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--envs", type=int, default=None, help="scenes per GPU (default 4096 for --scaling weak, 4096 / gpus for strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 4096 scenes per GPU; strong: 4096 scenes in total, 4096 / N per GPU (SURVEY.md section 8e)")
    ap.add_argument("--rule", choices=("aimed", "uniform"), default="aimed", help="action rule of the timed rounds (see the module docstring)")
    ap.add_argument("--groups", type=int, default=None, help="(default: 4 at 4096 scenes per GPU and more, 2 below -- shards of 2048 / 1024 / 512 scenes: 14.3 / 7.6 / 4.2 M with four groups against 14.6 / 7.5 / 4.4 M with two) scene groups per GPU, one engine handle + HIP stream each, rounds pipelined (1 = one handle). Round 6: 4 -- with launches "
                    "queued back to back four groups beat two by 1 %% on the headline and 2.7 %% on the six-object rounds, same box (profiles/r06_ae_scene_groups.log); until the blocking read "
                    "in front of every launch was found, more groups than two only added such waits (round 5: -20 %%). Eight groups share hardware queues: 8.4 M")
    ap.add_argument("--fused-rounds", type=int, default=-1, help="headline workload: K consecutive rounds of a scene per launch, the aiming rule evaluated in the kernel, no lock "
                    "step between scenes (ur5_grasp_rounds_dev; per-scene results bit-identical to K lock-step rounds). 0 = one launch per round, re-aimed on the device by torch; "
                    "default: default_rounds_per_launch() -- 2 at 4096 scenes per GPU and more (the chip is full: short launches, short tails), 8 / 16 for shards of 1024+ / fewer scenes, where a launch's end is the only tail")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the uniform-rule figure and the it4 / many sub-results (N = 1 only)")
    ap.add_argument("--sub", choices=("it4", "many", "many4096", "dqn", "dqn2048"), default=None,
                    help="run ONLY this secondary measurement (N = 1) and print it as {name: result}: what the rocprofv3 passes of tools/gpu_evidence_extras.sh profile")
    ap.add_argument("--sub-scenes", type=int, default=None, help="with --sub: scene count instead of the sub-result's own (same-box A/Bs of engine builds)")
    ap.add_argument("--sub-rounds", type=int, default=None, help="with --sub: timed rounds")
    ap.add_argument("--sub-fused", type=int, default=-1, help="with --sub it4 / many / many4096: rounds per launch with the observation rendered and the rule evaluated inside the launch (0 = the round-5 shape: "
                    "stand-alone render + torch rule + one launch per round); default: the sub-result's own")
    ap.add_argument("--sub-groups", type=int, default=2, help="with --sub-scenes / --sub-rounds: scene groups (handles + streams) of the sub-result; with --sub dqn / dqn2048: pipelined scene groups of the agent (1 = the serial loop)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for 2 ranks on one device)")
    ap.add_argument("--collectives", action="store_true", help="N = 1 only: create a one-rank process group on --backend and issue every collective of an N-rank job "
                    "(barriers, the per-round all_gather of outcome records, the final reductions) -- how the RCCL path is executed on a one-GPU box; results are unchanged")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mujoco_rl_ur5_amd import sharding
    from mujoco_rl_ur5_amd.model import load_model

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL), exactly as the driver's launcher would
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                          "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    ndev = torch.cuda.device_count()
    dev_id = local_rank % max(1, ndev)                                            # >1 rank per device only in the single-GPU shard-invariance check
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if world > 1:
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
    elif args.collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group(args.backend, rank=0, world_size=1, **({"device_id": dev} if args.backend == "nccl" else {}))
        sharding.FORCE_COLLECTIVES = True

    # six-object rounds: the headline's protocol, 20 timed rounds (round 5: 4 -- two launches per group, of which the region's edge is a quarter: same box 4 rounds 12.2 M, 8 rounds 13.0,
    # 12 rounds 13.3, 20 rounds 13.7 M env-steps/s, profiles/r06_ac_it4_timed_rounds.log)
    subs = {"it4": lambda cpu: rendered_sub_result(torch, dist, sharding, dev, dev_id, "it4", 4096, 20, 1, cpu, 4, SUB_FUSED["it4"]),
            # 10 timed rounds (round 5: 2): a region ends with its last launches draining alone -- about 1 s of a 4.4 s round here -- and two rounds measured that edge more than the rate
            # (2 rounds 645 k, 4 rounds 680 k, 6 rounds 714-717 k, 20 rounds 742 k; 4096 piles: 2 rounds 694 k, 8 rounds 742 k -- profiles/r06_al_pile_long_regions.log)
            "many": lambda cpu: rendered_sub_result(torch, dist, sharding, dev, dev_id, "many", 2048, 10, 1, cpu, 2, SUB_FUSED["many"]),   # BASELINE configs[3]: 16384 piles on 8 GPUs = 2048 per GPU
            "many4096": lambda cpu: rendered_sub_result(torch, dist, sharding, dev, dev_id, "many", 4096, 4, 1, False, 2, SUB_FUSED["many4096"]),   # north_star: "a 4096-env synthetic pile" on one GPU
            # pipelined scene groups (agent.BatchedGraspAgent): +3 % at 512 piles (one pile per CU: the other group's CNN finds LDS), -2 % at 2048 (two piles per CU hold
            # 99.5 % of a CU's LDS: a CNN kernel only gets a CU in the launch's tail, and runs 3.7 x slower there) -- same-box A/B in profiles/r05_g_dqn_ab.log
            "dqn": lambda cpu: dqn_sub_result(torch, dev, dev_id, 512, 2, 1, 2),
            # the same loop at the per-GPU scene count of configs[3] / [4] (16384 piles on 8 GPUs): with 512 piles a round lasts as long as its longest scene (2 piles per CU,
            # every scene resident at once: the chip idles through the tail), with 2048 the tail amortises
            "dqn2048": lambda cpu: dqn_sub_result(torch, dev, dev_id, 2048, 2, 1, 1)}
    if args.sub:
        if args.sub in ("dqn", "dqn2048") and (args.sub_scenes or args.sub_rounds or args.sub_groups != 2):
            dflt = {"dqn": (512, 2), "dqn2048": (2048, 1)}[args.sub]
            print(json.dumps({args.sub: dqn_sub_result(torch, dev, dev_id, args.sub_scenes or dflt[0], args.sub_rounds or dflt[1], 1, args.sub_groups)}), flush=True)
            return
        if args.sub in ("it4", "many", "many4096") and (args.sub_scenes or args.sub_rounds or args.sub_groups != 2 or args.sub_fused >= 0):
            wl = "it4" if args.sub == "it4" else "many"
            dflt = {"it4": (4096, 20), "many": (2048, 10), "many4096": (4096, 4)}[args.sub]
            res = rendered_sub_result(torch, dist, sharding, dev, dev_id, wl, args.sub_scenes or dflt[0], args.sub_rounds or dflt[1], 1, False, args.sub_groups,
                                      SUB_FUSED[args.sub] if args.sub_fused < 0 else args.sub_fused)
        else:
            res = subs[args.sub](False)
        print(json.dumps({args.sub: res}), flush=True)
        return
    model = load_model("it1_4box")
    n_local = args.envs if args.envs else (4096 if args.scaling == "weak" else 4096 // world)
    n_total = n_local * world
    lo, hi = sharding.shard_range(n_total, rank, world)
    if args.groups is None:
        args.groups = 4 if n_local >= 4096 else 2
    rounds = args.warmup + args.steps
    job = Job(torch, dist, sharding, model, "it1", args.rule, n_local, n_total, lo, world, dev, dev_id, args.groups, rounds + 8)
    groups, G, n_g, run_rounds, timed = job.groups, job.G, job.n_g, job.run_rounds, job.timed

    if args.fused_rounds < 0:
        args.fused_rounds = default_rounds_per_launch(n_local)
    fused = args.fused_rounds if args.rule == "aimed" else 0
    run_rounds(0, args.warmup, None, fused)
    elapsed, c0, c1, kernel_ms, gathered = timed(args.warmup, rounds, None, fused)
    launches = max(1, job.timed_launches)                                          # engine launches of the timed region on this rank (all scene groups)
    assert gathered.shape[1] == 4 and gathered.shape[0] % (n_g * world) == 0
    reward = torch.cat([gr.reward[:rounds] for gr in groups], dim=1)
    steps_local = int((c1["total_steps"] - c0["total_steps"]).sum())
    per_round = reward[args.warmup:].double().mean(dim=1)
    elapsed_local = elapsed
    stats = torch.tensor([elapsed, float(steps_local), float(reward[args.warmup:].sum().item())], dtype=torch.float64, device=dev)
    if sharding.collectives_active():
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
            traffic = tj["hbm_bytes_per_env_step"] * steps_local / launches                  # per launch, like algorithmic_bytes_per_launch below
            traffic_src = (f"from profiles/ ({tj.get('source', 'hbm_traffic_latest.json')}), NOT measured in this run: (2 x FETCH_SIZE + WRITE_SIZE) per "
                           "env-step of the rocprofv3 PMC passes of this command x env-steps of an average timed launch")
        out = {
            "metric": f"env-steps/sec (+ grasp-attempts/sec), {n_total} parallel UR5 scenes on {world} MI355X",
            "value": steps_all / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "scenes_total": n_total, "scenes_per_gpu": n_local,
            "grasp_attempts_per_s": attempts / elapsed, "grasp_success_rate": succ_all / attempts,
            "env_steps_total": int(round(steps_all)), "grasp_successes_total": int(round(succ_all)),
            "outcome_records_gathered_last": int(gathered.shape[0]),                     # rows of the last all_gather on rank 0: world x (rounds in it) x scenes of a group
            "steady_state_env_steps_per_s": job.steady_state_env_steps_per_s if world == 1 else None,   # rank 0's launches 2 .. L, both groups overlapping (Job.timed); `value` is the driver protocol's
            "grasp_success_rate_per_round_rank0": [round(float(x), 4) for x in per_round.tolist()],
            "env_steps_per_attempt": steps_all / attempts,
            "newton_iters_per_step": float((c1["solver_iters"] - c0["solver_iters"]).sum()) / max(1, steps_local),
            "status_bits": int(np.bitwise_or.reduce(c1["status"] | c1["status_ended"])),   # incl. the episodes that ended inside the timed launches
            "scenes_flagged": int(((c1["status"] | c1["status_ended"]) != 0).sum()),
            "scene_group_streams_overlap_verified": job.streams_overlap_verified,        # the groups' HIP streams sit on different hardware queues: measured (streams.py)
            "config": {"workload": "BASELINE.json configs[1]: IT1 (UR5gripper_2_finger.xml robot + bins, 4 equal 4 cm boxes), physics only, fixed "
                                   "z = 0.91, lift + 500-step closing check; one grasp-attempt round per step, episodes of 4 rounds with reset_model "
                                   "(+ 1000 ms settle) for the quarter of the batch that starts an episode in the round",
                       "rule": ("aimed: current position of a box still on the pick plate, read from the state records on the device" if args.rule == "aimed"
                                else "uniform: uniformly drawn table pixel, rotation 0 (SURVEY.md 8d / Grasping_Agent_multidiscrete.py:266-280)"),
                       "scenes_per_gpu": n_local, "scenes_total": n_total, "scene_groups_per_gpu": G,
                       "rounds_per_launch": (f"{fused}: the rule is evaluated in the kernel and a scene runs {fused} consecutive rounds (attempt, and reset_model + settle where its episode "
                                             "ends) without waiting for the other scenes; per-scene results bit-identical to one launch per round (tests/test_grasp_rounds.py)") if fused else "1 (lock step: every scene waits for the round's slowest)",
                       "solver": "Newton (MuJoCo default; north_star says PGS, see DESIGN.md D1), "
                       f"tolerance 1e-10, iteration cap {model.opt['iterations']}", "timestep_s": model.opt["timestep"],
                       "parallelism": f"scenes sharded x{world}, 1 all_gather of 16 B outcome records per scene group and round"
                                      + (f"; --collectives: one-rank process group, backend {dist.get_backend()}, every collective issued" if world == 1 and sharding.collectives_active() else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "traffic_source": traffic_src, "kernel": "ur5_run_kernel<32>", "bytes_per_env_step": bytes_per_step,
                         "avg_launch_ms": kernel_ms / launches, "launches_per_round": launches / args.steps, "launch_concurrency": G, "rounds_per_launch": args.steps * G / launches,
                         "env_steps_per_launch": steps_local / launches,
                         "algorithmic_bytes_per_launch": bytes_per_step * steps_local / launches,
                         "kernel_ms_per_round": 1e3 * elapsed_local / args.steps, "env_steps_per_round": steps_local / args.steps,
                         "note": "algorithmic state bytes x env-steps of the timed rounds / their wall time on rank 0: one attempt + episode-reset "
                                 "launch per scene group and round, the groups' launches overlapping on their own HIP streams (avg_launch_ms = "
                                 "HIP-event duration of one launch; achieved = launch_concurrency x bytes per launch / avg_launch_ms up to the "
                                 "overlap at the region's ends). A scene stays in LDS for a whole launch, so the algorithmic figure is an accounting "
                                 "unit, not the traffic: the step is latency / VALU-issue bound (rocprofv3 SQ counters, profiles/r04_q_pmc.txt: the fp64 VALU pipe of a SIMD is busy 63 % of the time, 40 % of the lanes of an instruction active; DESIGN.md section 3)"},
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
    if rank == 0 and world == 1 and not args.no_extras:
        out["strong_scaling_points"] = strong_scaling_points(torch, dist, sharding, model, dev, dev_id, args)
    job.close()
    if rank == 0 and world == 1 and not args.no_extras:
        torch.cuda.synchronize()
        # the headline line must not depend on the secondary measurements: a failure there is reported in place of the sub-result
        for key in ("it4", "many", "many4096", "dqn", "dqn2048"):
            fn = (lambda k=key: subs[k](not args.no_cpu_baseline))
            try:
                out[key] = fn()
            except Exception as exc:  # noqa: BLE001
                out[key] = {"error": f"{type(exc).__name__}: {exc}"}
        # the secondary results once more as top-level scalars (a driver that keeps only scalar keys of the line keeps these)
        for key in ("it4", "many", "many4096", "dqn", "dqn2048"):
            for f in ("env_steps_per_s", "grasp_attempts_per_s", "roofline_frac", "grasp_success_rate", "newton_iters_per_step", "update_to_data"):
                if isinstance(out.get(key), dict) and f in out[key]:
                    out[f"{key}_{f}"] = out[key][f]
        for pt in out.get("strong_scaling_points", []):
            if "env_steps_per_s_per_gpu" in pt:
                out[f"strong_{pt['scenes_per_gpu']}_scenes_env_steps_per_s_per_gpu"] = pt["env_steps_per_s_per_gpu"]
        if "cpu_baseline" in out:
            out["cpu_env_steps_per_s"], out["cpu_cores"] = out["cpu_baseline"]["value"], out["cpu_baseline"]["cores"]
        out["roofline_frac"] = out["roofline"]["frac"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or sharding.collectives_active():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
