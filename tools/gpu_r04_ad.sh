#!/bin/bash
# round 4: pile kernel that fetches the old values of the trailing update together with the loads of the panel rows (one global round trip per level instead of two) against
# the previous build (tools/libur5sim_head.so = HEAD): same bits on 256 piles (settle + attempt), then same-box A/B of bench.py --sub many at 2048 piles, 2 timed rounds
mkdir -p gpurun_out/r04ad
timeout 600 python tools/gpu_many_bits.py tools/libur5sim_head.so mujoco_rl_ur5_amd/csrc/libur5sim.so 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04ad/many_bits.log
bash tools/gpu_ab_many.sh r04ad 2048 2 tools/libur5sim_head.so
