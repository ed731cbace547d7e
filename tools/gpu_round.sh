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
           UR5SIM_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --fused-rounds ${b:--1} --envs ${c:-4096} --groups ${d:-2} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('%-28s K=%s G=${d:-2} n=%5d  %.3f M env-steps/s  %.1f attempts/s  %.1f ms/round  avg launch %.1f ms  success %.3f  status %d' % ('$lib'.split('/')[-1], '${b:-4}', d['scenes_per_gpu'], d['value'] / 1e6, d['grasp_attempts_per_s'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['grasp_success_rate'], d['status_bits']))" | tee -a $OUT/headline.log ;;
    bench) ( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -3 $OUT/bench_full.err; cut -c1-400 $OUT/bench_full.json ;;
    sub) timeout 900 python bench.py --sub $a > $OUT/sub_$a.json 2> $OUT/sub_$a.err; cut -c1-600 $OUT/sub_$a.json ;;
    py) args="${b:-} ${c:-}"; timeout 900 python $a ${args//:/ } > $OUT/$(basename $a .py).log 2>&1; tail -5 $OUT/$(basename $a .py).log ;;
    *) echo "unknown step $step" ;;
  esac
done
