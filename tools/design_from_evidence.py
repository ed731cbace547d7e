#!/usr/bin/env python3
"""Round 6: rewrites the number-bearing parts of DESIGN.md (sections 0, 2.3, 2.4, 3, 4, 5) from the evidence files under profiles/r06_x_* -- the generator the final DESIGN.md came from.
One-off (it patches the round-5 text kept in /tmp/design/base.md, which is not in the repository); kept for the record of which file every number was read from."""
import json, re, os, sys
R='/root/repo/'
base=open('/tmp/design/base.md').read()
B=json.loads([l for l in open(R+'profiles/r06_x_bench_full.json') if l.startswith('{')][-1])
M=lambda v: '%.2f M' % (v/1e6)
K=lambda v: '%.0f k' % (v/1e3)
hv, st = B['value'], B['steady_state_env_steps_per_s']
it4, many, many4 = B['it4_env_steps_per_s'], B['many_env_steps_per_s'], B['many4096_env_steps_per_s']
s2048, s1024, s512 = (B['strong_%d_scenes_env_steps_per_s_per_gpu' % n] for n in (2048, 1024, 512))
hbm=json.load(open(R+'profiles/r06_x_hbm_traffic.json')); mhbm=json.load(open(R+'profiles/r06_x_many_hbm_traffic.json'))
pmc=open(R+'profiles/r06_x_pmc.txt').read(); msq=open(R+'profiles/r06_x_many_sq_counters.txt').read()
def grab(txt, pat, cast=float):
    m=re.search(pat, txt); return cast(m.group(1)) if m else None
s=base
s=s.replace('(state at the end of round 5)','(state at the end of round 6)')
s=s.replace("This file is the CURRENT state only. How it got here — every experiment, A/B and superseded number of rounds 1-4 — is `HISTORY.md`; round 5's changes are marked \"(r5)\".",
 "This file is the CURRENT state only. How it got here — every experiment, A/B and superseded number of rounds 1-4 — is `HISTORY.md` (§7: round 5, §8: round 6 in the order it happened); round 5's changes are marked \"(r5)\", round 6's \"(r6)\".")
# ---- section 0
a=s.index('## 0. State at a glance'); b=s.index('## 1. The path and its boundary')
cpu=B['cpu_baseline']; mc=B['many'].get('cpu_baseline',{})
dq=B['dqn']; dq2=B['dqn2048']
sec0=f'''## 0. State at a glance (one MI355X; the driver's command `python bench.py --steps 20 --warmup 5`, `profiles/r06_x_bench_full.json`; every number of this file is from ONE binary, evidence set `profiles/r06_x_*`)

| workload (BASELINE.json) | kernel | env-steps/s | grasp attempts/s | round 5 (driver) | parity evidence |
|---|---|---|---|---|---|
| configs[1]: 4096 IT1 scenes (headline) | `ur5_run_kernel<32,64>`, 8 scenes per CU, 4 scene groups, 2 rounds per launch | **{M(hv)}** (steady state, launches 2 … L − 1: {M(st)}) | {B['grasp_attempts_per_s']/1e3:.2f} k | 17.24 M | phase step counts + grasp bit == oracle; arm 1e-9 rel; objects 1e-7 m; fused rounds == lock-step rounds word for word; 64 scenes of the 4096-scene launch == the oracle |
| strong-scaling shards: 2048 / 1024 / 512 scenes per GPU (4096 on 2 / 4 / 8 GPUs), 8 / 8 / 16 rounds per launch | same | {s2048/1e6:.2f} / {s1024/1e6:.2f} / {s512/1e6:.2f} M per GPU | {s2048/2035/1e3:.1f} / {s1024/2035/1e3:.1f} / {s512/2035/1e3:.1f} k | 14.47 / 7.54 / 3.98 M | same tests |
| configs[2]: 4096 six-object scenes, each rendering its own 200×200 RGB-D frame per round inside the launch | `ur5_run_kernel<44,64>`, 7 scenes per CU, 4 scene groups, 2 rounds per launch | {M(it4)} | {B['it4_grasp_attempts_per_s']/1e3:.2f} k | 11.97 M (4-round region) | same tests on the six-object scene, untouched objects 1e-8 m; in-launch observation == lock-step rounds word for word |
| configs[3]: 2048 / 4096 40-object piles, rendered | `ur5m_run_kernel<248,256>`, 2 scenes per CU, 2 scene groups, lock-step launches | **{K(many)} / {K(many4)}** | {B['many_grasp_attempts_per_s']:.0f} / {B['many4096_grasp_attempts_per_s']:.0f} | 634 k / 674 k (2-round regions) | (r6) every contact bit-equal to the oracle's from the same state; trajectories part from the oracle when the oracle's independent-arithmetic twins do (§4); run-to-run deterministic; grasp bit at the oracle's own chaos floor; equal Newton iteration counts |
| configs[4] shape on one GPU: DQN loop, 512 / 2048 piles | pile kernel + CNN | {K(dq['env_steps_per_s'])} / {K(dq2['env_steps_per_s'])} | {dq['grasp_attempts_per_s']:.0f} / {dq2['grasp_attempts_per_s']:.0f} | 173 / 336 attempts/s | CNN vs vectors of the reference's own `Modules.py`; replay cadence vs the reference's `ReplayBuffer`; (r6) the loop LEARNS: greedy success 0.96 against 0.011 for random actions after 60 rounds on IT1 scenes (`profiles/r06_dqn_learning_curve.json`) |
| CPU oracle, {cpu['cores']} threads of the GPU box (one thread: {cpu['single_core']['value']/1e3:.1f} k) | `oracle/ur5_oracle.cpp` | {K(cpu['value'])} (IT1), {mc.get('value',0)/1e3:.1f} k (piles) | {cpu['grasp_attempts_per_s']:.0f} | — | — |

53 GPU tests + smoke, 133 CPU tests. What moved in round 6: the headline +6 % (17.24 → {hv/1e6:.2f} M, driver to driver) from the LAUNCH STRUCTURE, not the kernel — nothing but engine launches between two
launches of a stream, K = 2 (§3); the 512-scene shard of the metric's strong-scaling shape {s512/3.98e6:.2f} × (K = 16); six-object scenes and piles are timed over regions long enough to measure the rate instead of the region's edge (§3; piles also: narrow phase 160 k → 38 k cycles); and parity: the pile kernel now parts from the oracle no earlier than the oracle's own independent-arithmetic twins do (§4). Roofline by the contract's accounting (state bytes per
env-step ÷ 8 TB/s): {100*B['roofline_frac']:.2f} % headline, {100*B['many_roofline_frac']:.2f} % piles — the scenes live in LDS for a whole launch and the step is a chain of dependent latencies (§3); what binds each kernel is
stated there from the SQ counters.

'''
s=s[:a]+sec0+s[b:]
# ---- section 1
s=s.replace('a C ABI (`extern "C"`, plain pointers and sizes, 32 entries, each citing the reference line it replaces)','a C ABI (`extern "C"`, plain pointers and sizes, 34 entries, each citing the reference line it replaces)')
s=s.replace('Scheduling entry points (none changes a result): `ur5_set_stream`, `ur5_set_order_dev`, `ur5_grasp_attempt_reset_dev`, `ur5_grasp_rounds_dev`, `ur5_kernel_ms_total`.',
 'Scheduling entry points (none changes a result): `ur5_set_stream`, `ur5_set_order_dev`, (r6) `ur5_set_order_view_dev` (the order read in place), `ur5_grasp_attempt_reset_dev`, `ur5_grasp_rounds_dev`,\n(r6) `ur5_set_observation_dev` (the observation rendered by the scene inside the launch, §2.4), `ur5_kernel_ms_total`.')
s=s.replace('| the episode loop of a scripted policy — `example_agent.py:15-27` | `ur5_grasp_rounds_dev` (r5): K rounds per scene and launch, the policy evaluated in the kernel |',
 '| the episode loop of a scripted policy — `example_agent.py:15-27` | `ur5_grasp_rounds_dev` (r5): K rounds per scene and launch, the policy evaluated in the kernel; (r6) with `get_observation` (`GraspingEnv.py:390-406`) inside it: `ur5_set_observation_dev` |')
# ---- 2.2 addition
s=s.replace('* Not adoptable, measured (r5): a PARTIAL refactorisation.','''* (r6) **The pile unit's position-level arithmetic is the oracle's, to the bit** (`UR5_STRICT` regions: no contraction; quaternions normalised twice by division as `mj_kinematics`; box-box
  vertices by the oracle's clipping; `q + h v` in two roundings): kinematics and collision are discontinuous in qpos (portal refinement of cylinder pairs), so a last-bit difference there — not in
  the velocities — is what starts a divergence (§4). The dynamics keep their fused multiply-adds and the reciprocal-square-root Cholesky. Cost: within noise (`gpurun_out/r06_d/ab_many.log`: 629-639 k against 645 k for the round-5 kernel, one box).
* (r6) Narrow phase: the broad-phase survivors that need per-lane portal refinement are filed from the END of the candidate list, the analytic pairs from its front, so the wavefronts that run the serial
  fp64 MPR chains run little else: 145 k → 38 k cycles per grasp step (profile build, `gpurun_out/r06_d/phases_new.log`), bit-identical.
* Not adoptable, measured (r5): a PARTIAL refactorisation.''')
# ---- 2.3, 2.4
a=s.index('### 2.3 Residency and registers'); b=s.index('### 2.4 Observation'); c=s.index('## 3. Kernels, roofline')
s=s[:a]+open('/tmp/design/s23.txt').read()+'\n'+open('/tmp/design/s24.txt').read()+'\n'+s[c:]
# ---- section 3
a=s.index('Evidence sets: `profiles/r05_x_*`'); b=s.index('| kernel | contract roofline')
s=s[:a]+'''Evidence: `profiles/r06_x_*`, ONE gpurun call on the shipped `libur5sim.so` (`tools/gpu_final.sh`): the 53 GPU tests + smoke, the bench line, rocprofv3 kernel stats, instruction / SQ busy-wait
counters and HBM-traffic passes for BOTH the headline and the pile kernel, 3 072-pile determinism, 1 024 + 768 grasp agreements, contact bits, capped replays, the RCCL one-rank lines.

'''+s[b:]
a=s.index('| kernel | contract roofline'); b=s.index('The launch is a makespan problem')
valu=grab(pmc,r'SQ_INSTS_VALU per env-step: (\d+)',int); salu=grab(pmc,r'SQ_INSTS_SALU per env-step: (\d+)',int); lds=grab(pmc,r'SQ_INSTS_LDS per env-step: (\d+)',int); vrd=grab(pmc,r'SQ_INSTS_VMEM_RD per env-step: (\d+)',int); smem=grab(pmc,r'SQ_INSTS_SMEM per env-step: (\d+)',int)
lane=grab(pmc,r'VALU lane utilisation[^=]*=[^=]*= ([0-9.]+)'); busy=grab(pmc,r'VALU busy share of resident-wave time[^=]*=[^=]*= ([0-9.]+)')
mw=grab(msq,r'waiting share[^=]*=[^=]*= ([0-9.]+)'); mi=grab(msq,r'issuing = [^=]*= ([0-9.]+)'); mwi=grab(msq,r'waiting to issue = [^=]*= ([0-9.]+)'); mb=grab(msq,r'VALU busy share[^=]*=[^=]*= ([0-9.]+)'); ml=grab(msq,r'VALU lane utilisation[^=]*=[^=]*= ([0-9.]+)')
mv=grab(msq,r'SQ_INSTS_VALU = \S+\s+\(([0-9.]+) per'); ms_=grab(msq,r'SQ_INSTS_SALU = \S+\s+\(([0-9.]+) per'); mlds=grab(msq,r'SQ_INSTS_LDS = \S+\s+\(([0-9.]+) per'); mvr=grab(msq,r'SQ_INSTS_VMEM_RD = \S+\s+\(([0-9.]+) per'); mvw=grab(msq,r'SQ_INSTS_VMEM_WR = \S+\s+\(([0-9.]+) per')
tab=f'''| kernel | contract roofline (HBM 8 TB/s) | counter traffic per env-step | what binds (SQ counters, same binary) |
|---|---|---|---|
| `ur5_run_kernel<32,64>` | {M(hv)} × 2 288 B = {hv*2288/1e9:.1f} GB/s = **{100*hv*2288/8e12:.3f} %** | {hbm['hbm_bytes_per_env_step']:.0f} B = {hbm['ratio_to_algorithmic']:.2f} × algorithmic (`profiles/r06_x_hbm_traffic.json`: 99 % of it writes = register-spill write-back, 504 B of scratch per lane) | dependent-instruction latency: fp64 VALU pipe busy {100*2*busy:.0f} % per SIMD (2 waves × {100*busy:.0f} %), {100*lane:.0f} % of an instruction's lanes active (8 robot rows, ≤ 30 contacts, 32 dofs); {valu/1e3:.1f} k VALU + {salu/1e3:.1f} k SALU + {lds/1e3:.2f} k LDS + {vrd/1e3:.2f} k VMEM reads (lane-indexed MODEL reads through the vector cache — geom sizes / poses / hull vertices: 392 static global loads in collision, MPR, kinematics and rows — of which ≈ 45 are scratch reloads: 10 in the move loop, ≈ 35 in `step_fn`; round 5 booked all of them as scratch) + {smem/1e3:.2f} k scalar-memory wave-instructions per step (`profiles/r06_x_pmc.txt`). The kernel is at its layout's ceiling (LDS caps residency at 8 scenes per CU, a third wave per SIMD costs 26 %): round 6 gained from the launch structure below |
| `ur5_run_kernel<44,64>` | {M(it4)} × 2 896 B = {100*it4*2896/8e12:.2f} % | — | the same, at 7 scenes per CU (the eighth: built, measured, +2 %: §2.3); the in-launch render is 2.8 % of a round |
| `ur5m_run_kernel<248,256>` | {K(many)} × 13 232 B = {many*13232/1e9:.1f} GB/s = **{100*many*13232/8e12:.2f} %** (4096 piles: {100*many4*13232/8e12:.2f} %) | **{mhbm['hbm_bytes_per_env_step']/1e3:.1f} KB = {mhbm['traffic_over_algorithmic']:.2f} × algorithmic** (`profiles/r06_x_many_hbm_traffic.json`; round 5: 24 KB, round 4: 427 KB): {100*mhbm['write_bytes']/(mhbm['write_bytes']+mhbm['fetch_bytes_corrected']):.0f} % of it writes = the write-back of 880 B of register spills per lane (the strict regions of round 6 cost registers); factor, block cache and staging live in LDS | barriers and LDS round trips of ≈ 30 phases per Newton iteration × 10.6 iterations: a wavefront is parked at a barrier / `s_waitcnt` **{100*mw:.1f} %** of its cycles (round 5, earlier build: 64.4 %), issues {100*mi:.1f} %, waits to issue {100*mwi:.1f} %; VALU busy {100*mb:.1f} % per wave ({100*2*mb:.0f} % per SIMD), {100*ml:.1f} % of an instruction's lanes active; {mv/1e3:.0f} k VALU + {ms_/1e3:.0f} k SALU + {mlds/1e3:.0f} k LDS + {(mvr+mvw)/1e3:.1f} k VMEM wave-instructions per env-step (`profiles/r06_x_many_sq_counters.txt`). Per grasp step 1.72 M cycles (profile build): factorisation 328 k + Hessian assembly 299 k + gradient / gather 163 k + sweeps 163 k + line search 152 k (wavefront 0 alone: round 4 measured the spread-out version as slower, HISTORY §8) + rows 84 k + images 74 k + narrow phase 38 k (r6; 160 k before) + the rest |
| `ur5_render_kernel` | 261 GB/s = 3.3 % | = algorithmic | ray-shape tests (fp32 VALU) |

'''
s=s[:a]+tab+s[b:]
a=s.index('The launch is a makespan problem'); b=s.index('## 4. Oracle')
s=s[:a]+open('/tmp/design/s3_makespan.txt').read()+'\n'+s[b:]
# ---- section 4 divergence bullet
a=s.index('* (r5) **When they part**'); b=s.index('* Not resolved: `media/console.png`')
div=open('/tmp/design/s4_div.txt').read()
dj=R+'profiles/r06_pile_divergence_time.json'
if os.path.exists(dj):
    D=json.load(open(dj)); S=D['summary']
    txt=(f"Result, 256 attempts: kernel vs oracle median **{S['kernel_median_steps_to_divergence']:.0f} steps** (quartiles {S['kernel_vs_oracle_quartiles'][0]:.0f}-{S['kernel_vs_oracle_quartiles'][1]:.0f}); the control twin "
         f"(fused dynamics, strict geometry) vs the oracle {S['control_twin_fused_dynamics_strict_geometry_median_steps']:.0f} ({S['control_twin_quartiles'][0]:.0f}-{S['control_twin_quartiles'][1]:.0f}), reciprocal-square-root Cholesky {S['control_twin_rsqrt_cholesky_median_steps']:.0f}, "
         f"everything fused (geometry too) {S['control_twin_fused_everywhere_median_steps']:.0f}; summation-order / 1-ulp twins {S['summation_order_and_ulp_twins_median_steps']:.0f}, elimination-order twin {S['elimination_order_twin_median_steps']:.0f}. "
         f"Kernel / control = {S['kernel_over_control_twin']:.2f} (criterion ≥ 0.9: {'met' if S['kernel_parts_no_earlier_than_0_9_x_the_control_twin'] else 'NOT met'}; ≥ 72 steps: {'met' if S['kernel_median_at_least_72_steps'] else 'NOT met'})")
else:
    txt="Result: PENDING (the oracle side is being computed)"
s=s[:a]+div.replace('**@DIV@**',txt)+s[b:]
s=s.replace("* **Piles are chaotic, measured**: the oracle agrees with its own rounding-level twins (contact list reversed; one coordinate + 1 ulp) on 96.1-97.3 % of the grasp bits of 256 selected\n  piles; the kernel agrees with the three on 95.7-97.3 % (`profiles/r04_q_pile_chaos_floor_*`; the r5 kernel is bit-identical to that kernel).",
 "* **Piles are chaotic, measured** (re-run in round 6 because the kernel's bits changed; 256 selected piles from the shipped kernel's settled states, `profiles/r06_x_pile_chaos_floor_*.json`): the oracle agrees with its\n  summation-order / 1-ulp twins on 97.3-98.0 % of the grasp bits (result codes 92.6-94.1 %), with its independent-arithmetic control twins (fused dynamics + strict geometry; reciprocal-square-root Cholesky) on 96.9-98.4 % (codes 91.0-92.2 %);\n  the kernel agrees with the oracle and those five on 95.3-96.9 % (codes 90.2-92.6 %): at the control twins' floor (96.9 % against 96.9 % for the pair that shares the kernel's arithmetic split), 1-2 points = 3-5 scenes of 256 (≈ 1.5 σ) below the twins that\n  share every instruction but one with the oracle. `tests/test_many_objects.py` holds 24 attempts on the GPU to that floor.")
s=s.replace('(r5) 2 × 1024 IT1 + 2 × 768 six-object attempts on the FINAL kernels: `profiles/r05_y_grasp_agreement_*.json`.','2 × 1024 IT1 + 2 × 768 six-object attempts on the shipped kernels, grasp bits and the 12 phase step counts 100 %: `profiles/r06_x_grasp_agreement_*.json`; (r6) the same at the full BASELINE size, 2 × 4096 + 2 × 4096 attempts: grasp bits 100 % in all four runs, the 12 phase step counts identical in 100 / 100 / 100 / 99.95 % of the scenes (2 of 4096 six-object scenes, `check_mode` 1), arm ≤ 1.4e-6 relative (`r06_x_grasp_agreement_4096_*.json`); a sampled 64 of the 4096-scene launch against the oracle, untouched objects bounded at 1e-8 m in the six-object tests.')
open(R+'DESIGN.md','w').write(s)
print('written', len(s.splitlines()), 'lines')
# ================= part 2: sections 2.2 refs, 5, 6, 7
s=open(R+'DESIGN.md').read()
det=json.load(open(R+'profiles/r06_x_many_determinism_3072piles.json'))
s=s.replace('3 072 piles in `profiles/r05_y_many_determinism_3072piles.json`)','3 072 piles in `profiles/r06_x_many_determinism_3072piles.json`)')
s=s.replace('| D11 | 96 contact slots per pile (reference: 1 500 = unbounded); 3 072 settled + grasped piles peak at 84 (4 of them above 80); overflow is flagged and surfaced by `GraspEnv` | `profiles/r05_y_many_determinism_3072piles.json` |',
 f"| D11 | 96 contact slots per pile (reference: 1 500 = unbounded); 3 072 settled + grasped piles peak at {det['ncon_max_max']} on the round-6 kernel (84 on round 5's); overflow is flagged and surfaced by `GraspEnv`; in the bench's 11 rounds of 2048 piles {B['many'].get('scenes_flagged', '?')} scene(s) exceed the slots at some step (`many.scenes_flagged`; the image has no byte left for more: two piles per CU) | `profiles/r06_x_many_determinism_3072piles.json` |")
s=s.replace('| (b) boundary | `include/ur5sim.h` (32 entries)','| (b) boundary | `include/ur5sim.h` (34 entries)')
a=s.index('| (d) measurement |'); b=s.index('| (f3), (f4) |')
cj=R+'profiles/r06_x_dqn_collectives_nccl.json'
coll=''
if os.path.exists(cj):
    try:
        C=json.load(open(cj))['dqn']; cp=C.get('collectives_per_round') or {}; est=C.get('collectives_xgmi_estimate_ms_per_round_8_gpus') or {}
        parts=[f"{k}: {v['calls']:.1f} calls, {v['mbytes']:.1f} MB, {v['device_ms']:.1f} ms" for k,v in cp.items()]
        coll=(" (r6) **What the agent path's collectives cost**, issued by one rank through RCCL (`bench.py --sub dqn --collectives --backend nccl`, `profiles/r06_x_dqn_collectives_nccl.json`), per round of 512 piles: "
              + "; ".join(parts) + f" of device time; ring estimate on one 153 GB/s xGMI link for 8 GPUs: broadcast {est.get('broadcast',0):.2f} ms + all-reduce {est.get('all_reduce',0):.2f} ms per round, against a round of {C['ms_per_round']:.0f} ms.")
    except Exception as e:
        coll=f" (collective timing: unreadable, {e})"
rows=f'''| (d) measurement | `bench.py`: headline + `uniform_rule`, `it4`, `many`, `many4096`, `dqn`, `dqn2048`, `strong_scaling_points`, each also as top-level scalars; `roofline` (live HIP events) + `cpu_baseline` (same workload; the pile leg aims with the GPU rounds' box rule and (r6) lists the GPU's round-0 rewards of the same scene ids); (r6) `steady_state_env_steps_per_s`, `env_steps_total`, `grasp_successes_total`; rocprofv3 stats / PMC / HBM passes of BOTH kernels on the shipped binary under `profiles/r06_x_*` (`tools/gpu_final.sh`); `roofline.traffic` is still read from `profiles/hbm_traffic_latest.json` (= `r06_x_hbm_traffic.json`, stated in `traffic_source`: PMC passes cannot run inside the timed region) |
| (e) multi-GPU | `sharding.py`: contiguous scene ranges, global-id seeds, ONE 16 B/scene `all_gather` per launch region; 1 rank ≡ 2 gloo ranks bit for bit (CPU and on one MI355X). (r6) the WHOLE driver path with two ranks on one device — `bench.py --gpus 2 --backend gloo`, weak and strong — against the one-rank run: `n_gpus`, `scenes_total`, gathered records, summed env-steps (`tests/test_sharding.py`, `-m gpu`). **Strong scaling is the metric's shape** ("4096 scenes on 1/2/4/8 MI355X"): `bench.py --scaling strong`; the measured per-GPU shard rates (K = 8 / 8 / 16 rounds per launch) put 4096 scenes on 2 / 4 / 8 GPUs at {2*s2048/1e6:.1f} / {4*s1024/1e6:.1f} / {8*s512/1e6:.1f} M env-steps/s against {hv/1e6:.2f} M on one — × {8*s512/hv:.2f} on 8 GPUs (round 5: × 1.85, round 4: × 1.36); what is left is the launch-boundary tail of the slowest scene and the lone-wavefront step time (512 scenes are 2 waves per CU); K = 16 costs outcome latency: a record is gathered up to 16 rounds after its attempt. Weak scaling (4096 per GPU, the driver's default) has no data-path collective. **RCCL has executed this code** with one rank (`sharding.FORCE_COLLECTIVES`): init, `all_gather_into_tensor` on int32 `[n,4]`, device / host-detour / flattened broadcasts, the replay batch's all-reduce (`tests/test_sharding.py::test_rccl_collectives_execute_on_one_gpu`, `profiles/r06_x_bench_collectives_nccl.json`).{coll} No curve is claimed: one GPU |
| (f1) grasp-Q CNN in the loop | `qnet.py`, `agent.py`: GEMM conv paths, per-image batch-norm for action selection (D12). `pipeline_groups=G` (r5): one env + stream per scene group, group g+1's render / CNN / selection queued under group g's grasp launch, pushes + optimiser steps under the next launch — the same transitions, losses and weights as the serial loop (`tests/test_agent.py`); (r6) the learner waits on an event of every group's action selection before its first optimiser step of a round (advisor finding: the weights could be overwritten under a forward; GPU test with an inflated CNN time). Measured (`profiles/r05_g_*`): a wash (+3 % at 512 piles, −2 % at 2048: two resident piles hold 99.5 % of a CU's LDS) |
| (f2) replay + learner | `qnet.ReplayBuffer` (+ shared ring across ranks), `agent.Learner`; (r6) **it learns**: 512 IT1 scenes, 60 rounds, ε 1 → 0.2, ring 2 048: greedy success in the last ten rounds 0.79 / 0.96 / 0.96 with 16 / 64 / 256 optimiser steps per round (`update_to_data` 0.03 / 0.12 / 0.5) against 0.011 for random actions; @PILE_LEARN@ (`tools/gpu_dqn_learning.py`, `profiles/r06_dqn_learning_curve.json`). `BatchedGraspAgent.save()` writes the reference trainer's checkpoint dict (`Grasping_Agent_multidiscrete.py:560-575`), `load_path` resumes optimiser, step, ε and rotation counters (`tests/test_agent.py`). The bench line's `dqn` stays at 16 optimiser steps per round (a throughput line; the cadence is a constructor argument) |
'''
s=s[:a]+rows+s[b:]
a=s.index('Round-4 verdict targets against the final bench line'); b=s.index('## 6. Deviations from the reference')
tg=f'''Round-5 verdict items against the final state (`profiles/r06_x_*`): (1) pile divergence — contacts bit-equal, control twins added, kernel median vs control in §4, tests tightened: **done**;
(2) one evidence set on one binary, six-object object-joint bounds, 64-of-4096 oracle sample, GPU rewards beside the pile CPU leg: **done**; (3) pile kernel ≥ 720 k / 760 k — **{K(many)} / {K(many4)}** over 10 / 4 timed rounds, 742 k sustained at both sizes (20 / 8 rounds)
(720 k met, 760 k missed: narrow phase 160 k → 38 k cycles and the measurement's edge removed; the line search stays on one wavefront for the reason measured in round 4; the "29 % outside the launch" was the region's edge, §3);
parked share ≤ 55 % — **{100*mw:.1f} %** (missed); (4) headline ≥ 18.2 M with no torch kernel between launches — **{M(hv)}** (met; a blocking one-element read before every launch found in the trace of this very
evidence set and removed, `profiles/r06_x_headline_trace_finding.txt`); K = 8 no longer slower than K = 4 — it still is (16.96 against 17.74 M): short launches win on a full chip, §3; scratch reloads < 300 — they ARE: ≈ 45 per step
(the 446 VMEM reads per step are lane-indexed model reads; the ISA has 392 static global loads in the per-step functions and 10 + 31 scratch loads on the per-step path); (5) the loop learns, checkpoints saved: **done** (IT1: 0.96 against 0.011); (6) six-object scenes ≥ 12.5 M —
**{M(it4)}** on the headline's protocol (20 timed rounds; 12.2 M over round 5's 4-round region, which measured its own edge: §3) — met; the eighth scene per CU was BUILT and measured at +2 % (§2.3): closed by measurement; (7) two-rank driver test, collective timing: **done**; (8) render under overlap — the observation
moved INTO the launch (§2.4) instead of onto a priority stream (priority measured: −1 … −7 %, `profiles/r06_o_many_rounds_per_launch.log`); (9) 512-scene shard ≥ 4.4 M — **{M(s512)}** (K = 16).

'''
s=s[:a]+tg+s[b:]
s=s.replace("The reference's agent scripts beyond the loop shape (TensorBoard, checkpoint saving — `load_path` is honoured), plotting","The reference's agent scripts beyond the loop shape (TensorBoard; checkpoints ARE written and read since r6), plotting")
pl=R+'profiles/r06_dqn_learning_curve.json'
L=json.load(open(pl))['runs']
if 'many_128_long' in L:
    m=L['many_128_long']['summary']
    ptxt=f"512 piles: 40 rounds / 20 k transitions 0.020 against 0.017 (no signal); {m['rounds']} rounds with {m['max_updates_per_round']} steps per round and a ring of {m['mem_size']}: greedy {m['greedy_success_last_10_rounds']:.3f} against {m['random_action_success_all_rounds']:.3f} for random actions (× {m['greedy_last_10_over_random']:.1f})"
else:
    ptxt="512 piles, 40 rounds / 20 k transitions: 0.020 against 0.017 — no signal yet at 2 % positives"
s=s.replace('@PILE_LEARN@', ptxt)
open(R+'DESIGN.md','w').write(s)
print('part 2 written', len(s.splitlines()), 'lines')
