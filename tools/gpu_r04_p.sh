#!/bin/bash
mkdir -p gpurun_out/r04p
{
python tools/gpu_drop_parity.py mujoco_rl_ur5_amd/csrc/libur5sim.so 32
python tools/gpu_drop_parity.py tools/libur5sim_strict.so 32
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04p/drop_parity.log
timeout 600 tools/gpu_ab_libs.sh r04p tools/libur5sim_strict.so
timeout 600 python -m pytest tests/test_sharding.py -m gpu -x -q -k one_agent 2>&1 | tail -2
