#!/bin/bash
# Round 3, first GPU session: GPU parity suite on the full-hull assets, per-phase cycles, same-box A/B of the queued build options.
REPO=$(cd "$(dirname "$0")/.." && pwd); cd $REPO
OUT=gpurun_out/r03a; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest_gpu.log
timeout 200 python tools/gpu_profile_phases.py 4096 > $OUT/phase_cycles_4096.log 2>&1
timeout 100 python tools/gpu_profile_phases.py 64 > $OUT/phase_cycles_64.log 2>&1
bash tools/gpu_ab_list.sh r03a mujoco_rl_ur5_amd/csrc/libur5sim.so tools/libur5sim_dppc.so tools/libur5sim_opaque.so tools/libur5sim_supk8.so tools/libur5sim_supk2.so mujoco_rl_ur5_amd/csrc/libur5sim.so
tail -5 $OUT/pytest_gpu.log; cat $OUT/phase_cycles_4096.log | tail -30
