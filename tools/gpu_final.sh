#!/bin/bash
# Round-end evidence in ONE gpurun call: rocprofv3 stats + PMC passes of the bench command, then (with the fresh traffic figure in place) the
# full default bench line, the GPU test suite and smoke(). usage: tools/gpu_final.sh <tag> <prefix>
TAG=${1:-final}; PFX=${2:-r02_t}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
bash $REPO/tools/gpu_evidence.sh $TAG $PFX > $OUT/evidence.log 2>&1
if [ -s $OUT/${PFX}_hbm_traffic.json ]; then python - <<PY
import json
t = json.load(open("$OUT/${PFX}_hbm_traffic.json")); t["source"] = "profiles/${PFX}_hbm_traffic.json"
json.dump(t, open("$REPO/profiles/hbm_traffic_latest.json", "w"), indent=1); json.dump(t, open("$OUT/hbm_traffic_latest.json", "w"), indent=1)
PY
fi
cd $REPO
timeout 400 python bench.py > $OUT/${PFX}_bench_full.json 2> $OUT/bench_full.err
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/${PFX}_pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${PFX}_smoke.log 2>&1
tail -3 $OUT/${PFX}_pytest_gpu.log; cat $OUT/${PFX}_smoke.log | tail -2; head -c 600 $OUT/${PFX}_bench_full.json
