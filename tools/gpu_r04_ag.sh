#!/bin/bash
# round 4, last GPU call: (1) the solve sweeps share a pass's panels among the four wavefronts too (tree) against factorisation-only sharing (tools/libur5sim_many_fshare.so)
# and HEAD~ (tools/libur5sim_head.so): same bits on 256 piles, same-box A/B at 2048 piles; (2) pile GPU tests + the pile sub-results of bench.py on the tree's library
mkdir -p gpurun_out/r04ag
timeout 600 python tools/gpu_many_bits.py tools/libur5sim_head.so mujoco_rl_ur5_amd/csrc/libur5sim.so 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04ag/many_bits.log
bash tools/gpu_ab_many.sh r04ag 2048 2 tools/libur5sim_many_fshare.so tools/libur5sim_head.so
timeout 600 python -m pytest tests -q -m gpu -k "many or pile" 2>&1 | tail -3 | tee gpurun_out/r04ag/pytest_many.log
for sub in many4096 dqn; do timeout 400 python bench.py --sub $sub > gpurun_out/r04ag/r04_ag_${sub}_bench.json 2>/dev/null; cut -c1-300 gpurun_out/r04ag/r04_ag_${sub}_bench.json; done
