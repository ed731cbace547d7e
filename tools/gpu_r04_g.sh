#!/bin/bash
mkdir -p gpurun_out
{
for n in 512 1024 2048; do
python tools/gpu_residency_probe2.py mujoco_rl_ur5_amd/csrc/libur5sim.so $n
python tools/gpu_residency_probe2.py tools/libur5sim_prelean.so $n
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_g_residency_probe.log
