#!/bin/bash
# PREPARED, NOT RUN (round 4 ended without GPU minutes): pile kernel with 16- / 32-lane panel slots (-DUR5_PANEL_SLOT, csrc/ur5_engine.h) against the tree's library.
# Build first, here:   make -C mujoco_rl_ur5_amd/csrc variant_many NAME=slot16 EXTRA=-DUR5_PANEL_SLOT=16 ; make -C mujoco_rl_ur5_amd/csrc variant_many NAME=slot32 EXTRA=-DUR5_PANEL_SLOT=32
# then:                gpurun --timeout 900 -- 'bash tools/gpu_next_panel_slots.sh'
# Expected from tools/pile_structure_stats.py (4.7 -> 3.7 passes per factorisation): a few per cent; same bits (checked on the device-code-on-host build).
mkdir -p gpurun_out/next_slots
for l in tools/libur5sim_many_slot16.so tools/libur5sim_many_slot32.so; do
  timeout 300 python tools/gpu_many_bits.py mujoco_rl_ur5_amd/csrc/libur5sim.so $l 128 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/next_slots/many_bits.log
done
bash tools/gpu_ab_many.sh next_slots 2048 2 tools/libur5sim_many_slot16.so tools/libur5sim_many_slot32.so
