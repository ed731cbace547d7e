#!/bin/bash
# rocprofv3 evidence for the kernels the headline command does not run: the many-object kernel (bench.py --sub many), the render kernels (it4 / many rounds)
# and the CNN of the DQN loop (--sub dqn). Kernel stats for all three, HBM counters (separate passes, as the MI355X guide prescribes) for `many`.
# usage: tools/gpu_evidence_extras.sh <tag> <prefix>   ->  gpurun_out/<tag>/<prefix>_{many,it4,dqn}_kernel_stats.csv, <prefix>_many_hbm_traffic.json
set -u
TAG=${1:-evx}; PFX=${2:-r03}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for sub in many it4 dqn; do
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$sub -o r -- python $REPO/bench.py --sub $sub > $OUT/${PFX}_${sub}_bench.json 2> $OUT/stats_$sub.err
  f=$(find $OUT/stats_$sub -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 $f > $OUT/${PFX}_${sub}_kernel_stats.csv
done
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_many_$grp -o r -- python $REPO/bench.py --sub many > $OUT/pmc_many_$grp.json 2> $OUT/pmc_many_$grp.err
done
python - <<PY
import csv, glob, json, os
out = "$OUT"
tot = {}
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    s = 0.0
    for p in glob.glob(os.path.join(out, "pmc_many_" + grp, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p, newline="")):
            if "ur5m_run_kernel" in r["Kernel_Name"] or "ur5_run_kernel" in r["Kernel_Name"]:
                s += float(r["Counter_Value"])
    tot[grp] = s
try:
    b = json.loads([l for l in open(os.path.join(out, "pmc_many_FETCH_SIZE.json")) if l.startswith("{")][-1])["many"]
    steps_all = b["env_steps_per_s"] * b["ms_per_round"] * 1e-3 * (b["rounds"] + b["warmup"])   # timed + warm-up rounds (the initial settle adds 500 steps per pile)
    steps_all += 500 * b["scenes"]
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of 1 KB? -> MI355X_MICROARCH.md: values are in kilobytes; gfx950: FETCH_SIZE under-counts by 2 (64 B requests counted as 32 B)
    rd, wr = 2 * tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
    json.dump({"kernel": "ur5m_run_kernel<248,256>", "env_steps_all_launches": steps_all, "fetch_bytes_corrected": rd, "write_bytes": wr,
               "hbm_bytes_per_env_step": (rd + wr) / steps_all, "algorithmic_bytes_per_env_step": b["bytes_per_env_step"],
               "traffic_over_algorithmic": (rd + wr) / steps_all / b["bytes_per_env_step"],
               "note": "FETCH_SIZE x 2 (gfx950 note of the MI355X guide) + WRITE_SIZE, KiB units, summed over every engine launch of `bench.py --sub many` (separate PMC passes)"},
              open(os.path.join(out, "${PFX}_many_hbm_traffic.json"), "w"), indent=1)
except Exception as e:
    print("traffic summary failed:", e)
PY
ls $OUT
