#!/bin/bash
# One parameterised GPU session script (replaces the per-experiment tools/gpu_r04_*.sh): each STEP is a word, run in the order given; everything lands in gpurun_out/$TAG.
#   gpurun --timeout 900 -- 'bash tools/gpu_round.sh r05_a tests bits:tools/libur5sim_r04.so ab_many:2048:2:tools/libur5sim_r04.so,tools/libur5sim_many_x.so'
# steps:  tests                      pytest -m gpu (+ smoke)
#         bits:LIB[:N]               tools/gpu_many_bits.py LIB vs the tree's library on N piles (default 128): must be BIT-IDENTICAL
#         ab_many:N:R:LIB,LIB...     same-box A/B of pile-kernel builds, bench.py --sub many at N piles, R timed rounds (tree's library first and last)
#         ab_small:LIB,LIB...        same-box A/B of the headline kernel (bench.py timed rounds only)
#         ab_it4:LIB,LIB...          same for the six-object kernel (bench.py --sub it4)
#         headline:LIB:K:N[:G]       the headline rounds only (G scene groups, default 2) (no extras), library LIB ("tree" = the tree's), K rounds per launch (0 = lock step), N scenes -> one line in headline.log
#         bench                      the driver's command, full line -> bench_full.json
#         sub:NAME                   bench.py --sub NAME -> sub_NAME.json
#         pmc:SUB                    SQ counters of the engine kernel of `bench.py --sub SUB` (many | it4), one rocprofv3 --pmc pass per counter group -> SUB_sq_counters.txt
#         py:SCRIPT[:ARGS...]        python SCRIPT ARGS (':' separates arguments) -> SCRIPT's basename .log
set -u
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd); cd $REPO
OUT=gpurun_out/$TAG; mkdir -p $OUT
for step in "$@"; do
  IFS=: read -r what a b c d <<< "$step"
  echo "=== $step"
  case $what in
    tests) timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
           timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
    bits) timeout 600 python tools/gpu_many_bits.py $a mujoco_rl_ur5_amd/csrc/libur5sim.so ${b:-128} 2>&1 | grep -v amdgpu.ids | tee -a $OUT/many_bits.log | tail -2 ;;
    ab_many) bash tools/gpu_ab_many.sh $TAG $a $b ${c//,/ } ;;
    ab_small) bash tools/gpu_ab_libs.sh $TAG ${a//,/ } ;;
    ab_it4) bash tools/gpu_ab_it4.sh $TAG ${a//,/ } ;;
    headline) lib=$a; [ "$lib" = tree ] && lib=mujoco_rl_ur5_amd/csrc/libur5sim.so
           UR5SIM_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --fused-rounds ${b:--1} --envs ${c:-4096} --groups ${d:-2} ${BENCH_EXTRA:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('%-28s K=%s G=${d:-2} n=%5d  %.3f M env-steps/s  %.1f attempts/s  %.1f ms/round  avg launch %.1f ms  success %.3f  status %d  steady %.3f M' % ('$lib'.split('/')[-1], '${b:-4}', d['scenes_per_gpu'], d['value'] / 1e6, d['grasp_attempts_per_s'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['grasp_success_rate'], d['status_bits'], (d.get('steady_state_env_steps_per_s') or 0) / 1e6))" | tee -a $OUT/headline.log ;;
    bench) ( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -3 $OUT/bench_full.err; cut -c1-400 $OUT/bench_full.json ;;
    sub) timeout 900 python bench.py --sub $a > $OUT/sub_$a.json 2> $OUT/sub_$a.err; cut -c1-600 $OUT/sub_$a.json ;;
    pmc) i=0; ROOT=$(pwd); ( cd /tmp; export TMPDIR=/tmp
           for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
             i=$((i+1)); timeout 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_${a}_$i -o r -- python $ROOT/bench.py --sub $a > /tmp/pmc_${a}_$i.json 2> /tmp/pmc_${a}_$i.err
           done )
         python - "$a" "$OUT" <<'PY' | tee $OUT/${a}_sq_counters.txt
import csv, glob, json, os, sys
sub, out = sys.argv[1], sys.argv[2]
tot = {}
for p in glob.glob("/tmp/pmc_%s_*/**/*counter_collection.csv" % sub, recursive=True):
    for r in csv.DictReader(open(p, newline="")):
        if "run_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
b = json.loads([l for l in open("/tmp/pmc_%s_1.json" % sub) if l.startswith("{")][-1])[sub]
steps = b["env_steps_per_s"] * b["ms_per_round"] * 1e-3 * (b["rounds"] + b["warmup"]) + 491 * b["scenes"]      # timed + warm-up rounds + the initial settle
print("%s, bench.py --sub %s (%d scenes), all engine launches of the run: %.0f env-steps" % (b["kernel"], sub, b["scenes"], steps))
for k in sorted(tot): print("%s = %.4e   (%.1f per env-step)" % (k, tot[k], tot[k] / steps))
g = tot.get
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"): print("VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU"))))
if g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"): print("VALU busy share of resident-wave time = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"): print("waiting share (s_waitcnt / barriers) = SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f; waiting to issue = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f; issuing = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")))
PY
         ;;
    py) args="${b:-} ${c:-} ${d:-}"; timeout 900 python $a ${args//:/ } > $OUT/$(basename $a .py).log 2>&1; tail -5 $OUT/$(basename $a .py).log ;;
    *) echo "unknown step $step" ;;
  esac
done
