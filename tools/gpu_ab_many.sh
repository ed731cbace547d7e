#!/bin/bash
# Same-box A/B of engine builds on the 40-object pile: bench.py --sub many at n scenes (default 512: two waves of 256 CUs), r timed rounds after 1 warm-up,
# baseline (the tree's library) first and last.   tools/gpu_ab_many.sh tag n rounds lib...
tag=$1; n=${2:-512}; r=${3:-1}; shift; shift; shift
mkdir -p gpurun_out/$tag
run() { UR5SIM_LIB=$1 timeout 600 python bench.py --sub many --sub-scenes $n --sub-rounds $r 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['many']; print('%-36s %8.1f k env-steps/s  %7.1f ms kernel/round/group  %.2f Newton it/step  success %.3f  status %d' % ('$1'.split('/')[-1], d['env_steps_per_s'] / 1e3, d['kernel_ms_per_round_and_group'], d['newton_iters_per_step'], d['grasp_success_rate'], d['status_bits']))"; }
{
run mujoco_rl_ur5_amd/csrc/libur5sim.so
for l in "$@"; do run $l; done
run mujoco_rl_ur5_amd/csrc/libur5sim.so
} 2>&1 | tee gpurun_out/$tag/ab_many.log
