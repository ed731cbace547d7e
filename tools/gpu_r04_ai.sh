#!/bin/bash
# round 4, the last GPU seconds: trailing update over the folded lower triangle (tree) against HEAD (tools/libur5sim_head.so): same bits on 128 piles, A/B at 2048 piles
mkdir -p gpurun_out/r04ai
timeout 60 python tools/gpu_many_bits.py tools/libur5sim_head.so mujoco_rl_ur5_amd/csrc/libur5sim.so 128 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r04ai/many_bits.log
bash tools/gpu_ab_many.sh r04ai 2048 2 tools/libur5sim_head.so
