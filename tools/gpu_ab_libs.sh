#!/bin/bash
# Same-box A/B of engine builds: bench.py (timed rounds only) once per library, baseline first and last. Usage: tools/gpu_ab_libs.sh tag lib...
tag=$1; shift
mkdir -p gpurun_out/$tag
run() { UR5SIM_LIB=$1 timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('%-40s %.3f M env-steps/s  %.1f ms/round  success %.3f  status %d' % ('$1'.split('/')[-1], d['value'] / 1e6, d['ms_per_step'], d['grasp_success_rate'], d['status_bits']))"; }
{
run mujoco_rl_ur5_amd/csrc/libur5sim.so
for l in "$@"; do run $l; done
run mujoco_rl_ur5_amd/csrc/libur5sim.so
} | tee gpurun_out/$tag/ab.log
