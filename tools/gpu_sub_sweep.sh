#!/bin/bash
# same-box sweep of a rendered sub-result of bench.py over (scene groups, rounds per launch, group-0 stream priority): tools/gpu_sub_sweep.sh TAG SUB "G:K:P G:K:P ..." [scenes] [rounds]
TAG=$1; SUB=$2; CONF=$3; N=${4:-}; R=${5:-}
mkdir -p gpurun_out/$TAG
for c in $CONF; do
  IFS=: read -r g k p <<< "$c"
  UR5_GROUP0_HIGH_PRIORITY=$p timeout 900 python bench.py --sub $SUB --sub-groups $g --sub-fused $k ${N:+--sub-scenes $N} ${R:+--sub-rounds $R} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline())['$SUB']; print('$SUB groups=$g K=$k group0-high-priority=$p  scenes %d rounds %d  %.1f k env-steps/s  %.0f attempts/s  %.1f ms/round  kernel/round/group %.1f ms  success %.3f status %d' % (d['scenes'], d['rounds'], d['env_steps_per_s']/1e3, d['grasp_attempts_per_s'], d['ms_per_round'], d['kernel_ms_per_round_and_group'], d['grasp_success_rate'], d['status_bits']))" | tee -a gpurun_out/$TAG/sub_sweep.log
done
