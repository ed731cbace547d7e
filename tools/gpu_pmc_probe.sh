#!/bin/bash
# Instruction-cache / scalar-cache / LDS-wait counters of ONE grasp launch (tools/gpu_pcsample_run.py), one rocprofv3 --pmc pass per group.
tag=${1:-probe}; n=${2:-2048}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python $R/tools/gpu_pcsample_run.py $n > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); best = {}
rows = list(csv.DictReader(open(sys.argv[1])))
# the grasp launch = the ur5_run_kernel dispatch with the largest dispatch id
ids = sorted({int(r["Dispatch_Id"]) for r in rows if "ur5_run_kernel" in r["Kernel_Name"]})
last = ids[-1]
for r in rows:
    if int(r["Dispatch_Id"]) == last: acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items(): print("%-28s %.4e" % (k, v))
PY
done | tee $out/summary.txt
