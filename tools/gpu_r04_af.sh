#!/bin/bash
# round 4: pile kernel whose factorisation shares the panels of a pass among all four wavefronts (rows and row pairs of a panel split over the spare wavefronts) against
# the previous build (tools/libur5sim_head.so = HEAD): same bits on 256 piles (settle + attempt), then same-box A/B of bench.py --sub many at 2048 piles, 2 timed rounds
mkdir -p gpurun_out/r04af
timeout 600 python tools/gpu_many_bits.py tools/libur5sim_head.so mujoco_rl_ur5_amd/csrc/libur5sim.so 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04af/many_bits.log
bash tools/gpu_ab_many.sh r04af 2048 2 tools/libur5sim_head.so
