#!/bin/bash
timeout 600 python -m pytest tests/test_sharding.py -m gpu -x -q -k one_agent 2>&1 | grep -E "^E  |passed|failed" | head -30
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
