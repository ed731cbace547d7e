#!/bin/bash
# final evidence of a round (one gpurun call): GPU tests, smoke, the driver's bench command, rocprofv3 stats + PMC passes of headline / many / it4 / dqn, pile states for the CPU-side agreement run
set -u
TAG=${1:-final_r04}; PFX=${2:-r04_k}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${PFX}_pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/${PFX}_pytest_gpu.log; tail -3 $OUT/${PFX}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${PFX}_smoke.log 2>&1; tail -2 $OUT/${PFX}_smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/${PFX}_bench_full.json 2> $OUT/${PFX}_bench_full.err; tail -3 $OUT/${PFX}_bench_full.err; cut -c1-600 $OUT/${PFX}_bench_full.json
bash tools/gpu_evidence.sh $TAG/ev $PFX > $OUT/evidence.log 2>&1; tail -3 $OUT/evidence.log
bash tools/gpu_evidence_extras.sh $TAG/evx $PFX > $OUT/evidence_extras.log 2>&1; tail -3 $OUT/evidence_extras.log
timeout 600 python tools/gpu_many_dump.py 3072 256 $OUT/${PFX}_many_states.npz > $OUT/${PFX}_many_determinism_3072piles.json 2> $OUT/many_dump.err; cat $OUT/${PFX}_many_determinism_3072piles.json
