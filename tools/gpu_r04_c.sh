#!/bin/bash
mkdir -p gpurun_out
timeout 900 tools/gpu_ab_many.sh r04c 512 1 tools/libur5sim_r03.so tools/libur5sim_many_lsblock.so tools/libur5sim_many_globalenv.so
UR5_PROF_LIB=tools/libur5sim_prof.so timeout 600 python tools/gpu_profile_phases.py 256 many > gpurun_out/r04_c_many_phase_cycles_256piles.log 2>&1; tail -22 gpurun_out/r04_c_many_phase_cycles_256piles.log
timeout 300 python -m pytest tests/test_many_objects.py -m gpu -x -q 2>&1 | tail -3
