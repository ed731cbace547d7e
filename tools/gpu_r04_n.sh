#!/bin/bash
mkdir -p gpurun_out/r04n
timeout 600 python -m pytest tests/test_sharding.py -m gpu -x -q -k one_agent 2>&1 | grep -E "^E |passed|failed" | head -20
timeout 600 python -m pytest tests/test_dataset.py tests/test_qnet.py -m gpu -x -q 2>&1 | tail -2
timeout 600 tools/gpu_ab_libs.sh r04n tools/libur5sim_lsreg.so
