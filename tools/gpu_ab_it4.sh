#!/bin/bash
# Same-box A/B of engine builds on BOTH small-scene kernels: the headline rounds (ur5_run_kernel<32>) and the rendered six-object rounds (<44>, bench.py --sub it4).
# usage: tools/gpu_ab_it4.sh tag lib...
tag=$1; shift
mkdir -p gpurun_out/$tag
for l in "$@"; do
  a=$(UR5SIM_LIB=$l timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f M (status %d)' % (d['value']/1e6, d['status_bits']))")
  b=$(UR5SIM_LIB=$l timeout 300 python bench.py --sub it4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline())['it4']; print('%.3f M (success %.3f status %d)' % (d['env_steps_per_s']/1e6, d['grasp_success_rate'], d['status_bits']))")
  printf "%-28s headline %s   it4 %s\n" $(basename $l) "$a" "$b"
done | tee gpurun_out/$tag/ab.log
