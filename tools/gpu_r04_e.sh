#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_many_objects.py tests/test_constraint_rows.py -m gpu -x -q 2>&1 | tail -3
timeout 900 tools/gpu_ab_many.sh r04e 512 1 tools/libur5sim_prelean.so tools/libur5sim_r03.so
timeout 900 tools/gpu_ab_many.sh r04e2048 2048 1 tools/libur5sim_prelean.so
UR5_PROF_LIB=tools/libur5sim_prof.so timeout 600 python tools/gpu_profile_phases.py 512 many > gpurun_out/r04_e_many_phase_cycles_512piles.log 2>&1; tail -22 gpurun_out/r04_e_many_phase_cycles_512piles.log
