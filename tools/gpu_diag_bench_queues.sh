#!/bin/bash
# Which hardware queue does every engine launch of the FULL bench line land on? (rocprofv3 kernel trace of `bench.py`, engine kernels only.)  usage: tools/gpu_diag_bench_queues.sh TAG [lib...]
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  UR5SIM_LIB=$REPO/$lib timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$name -o r -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  f=$(find /tmp/trace_$name -name "*kernel_trace.csv" | head -1)
  head -1 $f > $OUT/trace_$name.csv; grep "ur5" $f >> $OUT/trace_$name.csv
  python - $OUT/bench_$name.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], "headline %.2f M" % (d["value"] / 1e6), {k: (round(d[k]["ms_per_round"], 1), round(d[k].get("kernel_ms_per_round_and_group", 0), 1)) for k in ("it4", "many", "many4096") if isinstance(d.get(k), dict)},
      [round(p["env_steps_per_s_per_gpu"] / 1e6, 2) for p in d.get("strong_scaling_points", [])])
PY
done
