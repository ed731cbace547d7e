#!/bin/bash
# round 4: Hessian-term staging in LDS chunks (-DUR5_STG_LDS) against the shipped global-scratch staging: throughput at 2048 piles and the L2 -> fabric traffic of the pile kernel
mkdir -p gpurun_out/r04u
REPO=$(pwd)
run() { UR5SIM_LIB=$1 timeout 600 python bench.py --sub many --sub-scenes 2048 --sub-rounds 2 --sub-groups 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['many']; print('%-28s %8.1f k env-steps/s  %7.1f attempts/s  %7.1f ms kernel/round/group  success %.3f status %d' % ('$1'.split('/')[-1], d['env_steps_per_s'] / 1e3, d['grasp_attempts_per_s'], d['kernel_ms_per_round_and_group'], d['grasp_success_rate'], d['status_bits']))"; }
{
run mujoco_rl_ur5_amd/csrc/libur5sim.so
run tools/libur5sim_many_stglds.so
run mujoco_rl_ur5_amd/csrc/libur5sim.so
} 2>&1 | tee gpurun_out/r04u/ab_many_stg_lds.log
cd /tmp; export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  UR5SIM_LIB=$REPO/tools/libur5sim_many_stglds.so timeout 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $REPO/gpurun_out/r04u/pmc_$grp -o r -- python $REPO/bench.py --sub many > $REPO/gpurun_out/r04u/pmc_$grp.json 2> $REPO/gpurun_out/r04u/pmc_$grp.err
done
cd $REPO
python - <<'PY'
import csv, glob, json, os
out = "gpurun_out/r04u"
tot = {}
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    s = 0.0
    for p in glob.glob(os.path.join(out, "pmc_" + grp, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p, newline="")):
            if "ur5m_run_kernel" in r["Kernel_Name"]:
                s += float(r["Counter_Value"])
    tot[grp] = s
b = json.loads([l for l in open(os.path.join(out, "pmc_FETCH_SIZE.json")) if l.startswith("{")][-1])["many"]
steps_all = b["env_steps_per_s"] * b["ms_per_round"] * 1e-3 * (b["rounds"] + b["warmup"]) + 500 * b["scenes"]
rd, wr = 2 * tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
res = {"kernel": "ur5m_run_kernel<248,256> built with -DUR5_STG_LDS", "env_steps_all_launches": steps_all, "fetch_bytes_corrected": rd, "write_bytes": wr,
       "hbm_bytes_per_env_step": (rd + wr) / steps_all, "algorithmic_bytes_per_env_step": b["bytes_per_env_step"], "traffic_over_algorithmic": (rd + wr) / steps_all / b["bytes_per_env_step"]}
json.dump(res, open(os.path.join(out, "many_hbm_traffic_stg_lds.json"), "w"), indent=1)
print(res)
PY
