#!/bin/bash
# round 4: SQ counters of the pile kernel (bench.py --sub many, 2048 piles, two scenes per CU), one rocprofv3 --pmc pass per group
REPO=$(pwd); OUT=$REPO/gpurun_out/r04x; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$i -o r -- python $REPO/bench.py --sub many > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err
done
cd $REPO
python - <<'PY'
import csv, glob, json, os
out = "gpurun_out/r04x"
tot = {}
for p in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p, newline="")):
        if "ur5m_run_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
b = json.loads([l for l in open(os.path.join(out, "pmc_1.json")) if l.startswith("{")][-1])["many"]
steps = b["env_steps_per_s"] * b["ms_per_round"] * 1e-3 * (b["rounds"] + b["warmup"]) + 500 * b["scenes"]
lines = ["ur5m_run_kernel<248,256>, bench.py --sub many (2048 piles, two scenes per CU), all engine launches of the run: %.0f env-steps" % steps]
for k in sorted(tot):
    lines.append("%s = %.4e   (%.1f per env-step)" % (k, tot[k], tot[k] / steps))
g = tot.get
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    lines.append("VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU"))))
if g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"):
    lines.append("VALU busy share of resident-wave time = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
    lines.append("waiting share (s_waitcnt / barriers) = SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f; waiting to issue = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")))
open(os.path.join(out, "r04_x_many_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
