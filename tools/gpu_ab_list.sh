#!/bin/bash
# Same-box A/B of engine builds given in order (repeat the baseline yourself): bench.py timed rounds only, one line per library.
# usage: tools/gpu_ab_list.sh tag lib...
tag=$1; shift
mkdir -p gpurun_out/$tag
for l in "$@"; do UR5SIM_LIB=$l timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('%-28s %.3f M env-steps/s  %.1f ms/round  success %.3f  status %d' % ('$l'.split('/')[-1], d['value'] / 1e6, d['ms_per_step'], d['grasp_success_rate'], d['status_bits']))"; done | tee gpurun_out/$tag/ab.log
