#!/bin/bash
# final evidence of a round (one gpurun call): GPU tests, smoke, the driver's bench command, rocprofv3 stats + PMC passes of headline / many / it4 / dqn, pile states for the CPU-side agreement run
set -u
TAG=${1:-final}; PFX=${2:-r06_x}
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
# round 5: the RCCL path on this one GPU (one-rank nccl process group, every collective issued), grasp agreement of the FINAL small-scene kernels against the oracle on the
# box's host cores, and the capped replays of the dumped piles for the CPU-side divergence-time statistic (tools/pile_divergence_time.py)
timeout 300 python bench.py --steps 8 --warmup 4 --no-extras --no-cpu-baseline --collectives --backend nccl > $OUT/${PFX}_bench_collectives_nccl.json 2> $OUT/bench_collectives.err; cut -c1-300 $OUT/${PFX}_bench_collectives_nccl.json
# the agent path's collectives (weights / Adam state broadcast, replay batch all-reduce, outcome all_gather) issued by ONE rank through RCCL: device ms per round in the dqn line
timeout 400 python bench.py --sub dqn --collectives --backend nccl 2> $OUT/dqn_collectives.err | grep "^{" | tail -1 > $OUT/${PFX}_dqn_collectives_nccl.json; python -c "
import json; d = json.load(open('$OUT/${PFX}_dqn_collectives_nccl.json'))['dqn']; print({k: d.get(k) for k in ('grasp_attempts_per_s', 'collectives_per_round', 'collectives_xgmi_estimate_ms_per_round_8_gpus')})"
timeout 600 python tools/gpu_agreement.py 1024 it1_4box 2>/dev/null | tail -1 > $OUT/${PFX}_grasp_agreement_1024_it1.json; cut -c1-400 $OUT/${PFX}_grasp_agreement_1024_it1.json
timeout 600 python tools/gpu_agreement.py 768 /UR5+gripper/UR5gripper_2_finger.xml 2>/dev/null | tail -1 > $OUT/${PFX}_grasp_agreement_768_2f.json; cut -c1-400 $OUT/${PFX}_grasp_agreement_768_2f.json
# round 6: the capped replays start from the FIXED states of tools/probes/states/ (dumped once from the round-5 kernel; the oracle side -- base + six twins x 256 attempts, hours of
# CPU -- is cached next to them and stays valid whatever the kernel's bits become); the SQ busy / wait counters of the PILE kernel on this same binary (round-5 verdict item 2)
ST=tools/probes/states/r06_many_states.npz; [ -f $ST ] || ST=$OUT/${PFX}_many_states.npz
timeout 300 python tools/gpu_many_divergence.py $ST $OUT/${PFX}_many_divergence_gpu.npz 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/${PFX}_many_divergence_gpu.json; cat $OUT/${PFX}_many_divergence_gpu.json
timeout 300 python tools/contact_bits.py 16 400 6 gpu 2>/dev/null | tail -1 > $OUT/${PFX}_contact_bits_gpu.json; cut -c1-200 $OUT/${PFX}_contact_bits_gpu.json
bash tools/gpu_round.sh $TAG pmc:many > $OUT/pmc_many.log 2>&1; cp $OUT/many_sq_counters.txt $OUT/${PFX}_many_sq_counters.txt 2>/dev/null; tail -4 $OUT/pmc_many.log
