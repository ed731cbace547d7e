#!/bin/bash
mkdir -p gpurun_out
timeout 900 tools/gpu_ab_many.sh r04d 512 1 tools/libur5sim_many_split512.so tools/libur5sim_many_occ2_split.so tools/libur5sim_many_occ2_flat.so
