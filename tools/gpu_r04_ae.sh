#!/bin/bash
# round 4: where the pile kernel's level loops spend their cycles (profile build with -DUR5_PROFILE_LEVELS, 512 piles = two scenes per CU), for the next round
mkdir -p gpurun_out/r04ae
UR5_PROFILE_LEVELS=1 UR5_PROF_LIB=tools/libur5sim_prof_levels.so timeout 600 python tools/gpu_profile_phases.py 512 many 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04ae/r04_ae_many_level_loop_cycles_512piles.log
