#!/bin/bash
mkdir -p gpurun_out/r04s
run() { UR5SIM_LIB=$1 timeout 600 python bench.py --sub many --sub-scenes $2 --sub-rounds $4 --sub-groups $3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['many']; print('%-28s scenes %5d groups %d  %8.1f k env-steps/s  %7.1f attempts/s  %7.1f ms kernel/round/group  %7.1f ms/round  success %.3f status %d' % ('$1'.split('/')[-1], $2, $3, d['env_steps_per_s'] / 1e3, d['grasp_attempts_per_s'], d['kernel_ms_per_round_and_group'], d['ms_per_round'], d['grasp_success_rate'], d['status_bits']))"; }
{
run mujoco_rl_ur5_amd/csrc/libur5sim.so 2048 2 2
run tools/libur5sim_c4b.so 2048 2 2
run mujoco_rl_ur5_amd/csrc/libur5sim.so 2048 2 2
} 2>&1 | tee gpurun_out/r04s/ab_many_coupling_run_accumulate.log
