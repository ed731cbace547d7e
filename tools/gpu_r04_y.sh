#!/bin/bash
# round 4: six-object kernel (NV = 44) with 7 scenes per CU (twists / search images / aref offsets in the Hessian's tail: 25 496 -> 23 088 B) against HEAD's 6,
# same box, bench.py --sub it4; then the six-object GPU parity tests on the new build and the headline rounds of both (the NV = 32 code is instruction-identical)
mkdir -p gpurun_out/r04y gpurun_out/r04z
timeout 120 tools/lds_residency_probe 8192 16384 19200 19201 20376 20480 20481 21760 21761 22744 23040 23041 23088 23376 24320 24321 25496 25600 25601 27276 28160 28161 32000 32001 40960 40961 54400 54401 81184 81920 81921 2>&1 | tee gpurun_out/r04z/lds_residency.log
for l in tools/libur5sim_head.so mujoco_rl_ur5_amd/csrc/libur5sim.so tools/libur5sim_head.so mujoco_rl_ur5_amd/csrc/libur5sim.so; do
  UR5SIM_LIB=$l timeout 300 python bench.py --sub it4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['it4']; print('%-24s it4 %.3f M env-steps/s  %.1f ms/round  success %.3f  status %s' % ('$l'.split('/')[-1], d['env_steps_per_s'] / 1e6, d['ms_per_round'], d.get('grasp_success_rate', -1), d.get('status_bits')))"
done | tee gpurun_out/r04y/ab_it4.log
timeout 900 python -m pytest tests -q -m gpu -k "six or it4 or two_finger or random_agent or failure or 2f" 2>&1 | tail -5 | tee gpurun_out/r04y/pytest_six.log
bash tools/gpu_ab_list.sh r04y tools/libur5sim_head.so mujoco_rl_ur5_amd/csrc/libur5sim.so
