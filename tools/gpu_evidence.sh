#!/bin/bash
# Evidence run of a round (one gpurun call): rocprofv3 kernel stats + PMC passes of the bench command, summaries under gpurun_out/<tag>/.
# usage: tools/gpu_evidence.sh <tag> [prefix]      (then copy gpurun_out/<tag>/<prefix>_* into profiles/; prefix defaults to r02_t)
set -u
TAG=${1:-ev}
PFX=${2:-r02_t}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
CMD="python $REPO/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline"
cd /tmp; export TMPDIR=/tmp
timeout 300 $CMD > $OUT/bench_plain.json 2> $OUT/bench_plain.err
[ "${EV_FAST:-0}" = 1 ] || timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- $CMD > $OUT/bench_stats.json 2> $OUT/stats.err   # EV_FAST=1: counters only (the stats pass was taken separately)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  d=$OUT/pmc_$(echo $grp | cut -d' ' -f1)
  [ "${EV_FAST:-0}" = 1 ] && [ "$(echo $grp | cut -d' ' -f1)" = SQ_THREAD_CYCLES_VALU ] && continue
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o r -- $CMD > $d.json 2> $d.err
done
python $REPO/tools/rocprof_summary.py $OUT $OUT/bench_plain.json $OUT/$PFX > $OUT/summary.json 2> $OUT/summary.err
ls $OUT
