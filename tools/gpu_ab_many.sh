#!/bin/bash
# Same-box A/B of engine builds on the 40-object pile: bench.py's "many" sub-result at n scenes (default 512: two waves of 256 CUs), 1 timed round after 1 warm-up.
tag=$1; n=$2; shift; shift
mkdir -p gpurun_out/$tag
run() { UR5SIM_LIB=$1 timeout 300 python - "$n" <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
r = bench.rendered_sub_result(torch, torch.device("cuda:0"), 0, "many", int(sys.argv[1]), 1, 1)
print("%-36s %8.1f k env-steps/s  %6.1f ms kernel/round  %.2f Newton it/step  success %.3f  status %d" % (os.environ["UR5SIM_LIB"].split("/")[-1], r["env_steps_per_s"] / 1e3, r["kernel_ms_per_round"], r["newton_iters_per_step"], r["grasp_success_rate"], r["status_bits"]))
PY
}
{
run mujoco_rl_ur5_amd/csrc/libur5sim.so
for l in "$@"; do run $l; done
run mujoco_rl_ur5_amd/csrc/libur5sim.so
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$tag/ab_many.log
