#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_many_objects.py tests/test_constraint_rows.py tests/test_sharding.py tests/test_agent.py tests/test_qnet.py -m gpu -x -q > gpurun_out/r04_b_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_b_pytest_gpu.log
tail -5 gpurun_out/r04_b_pytest_gpu.log
timeout 600 python tools/gpu_many_dump.py 3072 256 gpurun_out/r04_many_states.npz > gpurun_out/r04_b_many_dump.json 2> gpurun_out/r04_b_many_dump.err; cat gpurun_out/r04_b_many_dump.json
timeout 600 tools/gpu_ab_many.sh r04b 512 1 tools/libur5sim_r03.so
UR5_PROF_LIB=tools/libur5sim_prof.so timeout 600 python tools/gpu_profile_phases.py 256 many > gpurun_out/r04_b_many_phase_cycles_256piles.log 2>&1; tail -22 gpurun_out/r04_b_many_phase_cycles_256piles.log
