#!/bin/bash
mkdir -p gpurun_out/r04m
timeout 600 python -m pytest tests/test_sharding.py tests/test_dataset.py tests/test_qnet.py -m gpu -x -q 2>&1 | tail -3
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04m/r04_m_bench_full.json 2> gpurun_out/r04m/r04_m_bench_full.err; tail -3 gpurun_out/r04m/r04_m_bench_full.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04m/r04_m_bench_full.json') if l.startswith('{')][-1])
print({k: v for k, v in d.items() if isinstance(v, (int, float)) and not isinstance(v, bool)})
print(d.get('dqn2048'))
PY
