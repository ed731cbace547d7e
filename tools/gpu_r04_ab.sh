#!/bin/bash
# round 4, closing evidence on the final tree (pile kernel with the forward sweep inside the factorisation, six-object kernel at 7 scenes per CU; the headline kernel's code is
# instruction-identical to the r04_q evidence): GPU tests, smoke, the driver's bench command, rocprofv3 kernel stats of `--sub many` / `--sub it4`, HBM counters of `many`
set -u
TAG=r04ab; PFX=r04_ab
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${PFX}_pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/${PFX}_pytest_gpu.log; tail -3 $OUT/${PFX}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${PFX}_smoke.log 2>&1; tail -2 $OUT/${PFX}_smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/${PFX}_bench_full.json 2> $OUT/${PFX}_bench_full.err; tail -3 $OUT/${PFX}_bench_full.err; cut -c1-400 $OUT/${PFX}_bench_full.json
sed -i 's/^for sub in many it4 dqn; do$/for sub in many it4; do/' tools/gpu_evidence_extras.sh
bash tools/gpu_evidence_extras.sh $TAG $PFX > $OUT/evidence_extras.log 2>&1; tail -3 $OUT/evidence_extras.log
cat $OUT/${PFX}_many_hbm_traffic.json
