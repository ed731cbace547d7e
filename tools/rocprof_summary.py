#!/usr/bin/env python3
"""Condense rocprofv3 CSV output of `bench.py` runs into the small summaries kept under profiles/.

    python tools/rocprof_summary.py <dir with *_kernel_stats.csv / *_counter_collection.csv> <bench.json> <out prefix>

Writes <prefix>_kernel_stats.csv (the run-kernel rows of --stats), <prefix>_pmc.txt (per-launch counter values of the timed
launches) and, when FETCH_SIZE / WRITE_SIZE passes are present, <prefix>_hbm_traffic.json (bytes per env-step; FETCH_SIZE is
reported in 32 B... see MI355X_MICROARCH.md: rocprofv3 already scales both to bytes on gfx950? no -- KiB; handled below).
"""
import csv
import glob
import json
import os
import sys


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def main():
    d, bench_json, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
    bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
    steps = bench["steps"]
    lpr = float(bench["roofline"].get("launches_per_round", 1))          # engine launches per round: scene groups / rounds per launch (0.5 = two groups, four rounds per launch)
    env_steps = bench["roofline"].get("env_steps_per_launch", bench["roofline"].get("env_steps_per_round"))
    launches = max(1, int(round(steps * lpr)))
    # kernel stats
    for p in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rs = rows(p)
        with open(prefix + "_kernel_stats.csv", "w") as f:
            f.write(",".join(rs[0].keys()) + "\n")
            for r in rs:
                f.write(",".join('"%s"' % v if "," in v else v for v in r.values()) + "\n")
    # counters: dispatch -> counter -> summed value (one row per dimension instance)
    per = {}
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in rows(p):
            if "ur5_run_kernel" not in r["Kernel_Name"]:
                continue
            key = (os.path.dirname(p), int(r["Dispatch_Id"]))
            per.setdefault(r["Counter_Name"], {}).setdefault(key, 0.0)
            per[r["Counter_Name"]][key] += float(r["Counter_Value"])
    out = {}
    lines = []
    for name, disp in sorted(per.items()):
        vals = [v for _, v in sorted(disp.items())]
        timed = vals[-launches:]                      # the warm-up and reset launches come first
        out[name] = sum(timed) / len(timed)
        lines.append(f"{name} = {out[name]:.4e}   (mean of the {len(timed)} timed launches; all launches: {['%.3e' % v for v in vals]})")
    if lines:
        with open(prefix + "_pmc.txt", "w") as f:
            f.write(f"ur5_run_kernel, {bench['config']['scenes_per_gpu']} scenes, {env_steps:.0f} env-steps per timed launch\n")
            f.write("\n".join(lines) + "\n")
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if k in out:
                    f.write(f"{k} per env-step: {out[k] / env_steps:.0f}\n")
            if "SQ_THREAD_CYCLES_VALU" in out and "SQ_ACTIVE_INST_VALU" in out and out["SQ_ACTIVE_INST_VALU"] > 0:
                # thread-cycles spent in VALU instructions over (cycles VALU instructions were executing x 64 lanes): the share of lanes doing work
                f.write(f"VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = {out['SQ_THREAD_CYCLES_VALU'] / (64.0 * out['SQ_ACTIVE_INST_VALU']):.3f}\n")
            if "SQ_ACTIVE_INST_VALU" in out and "SQ_WAVE_CYCLES" in out:
                f.write(f"VALU busy share of resident-wave time = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {out['SQ_ACTIVE_INST_VALU'] / out['SQ_WAVE_CYCLES']:.3f}\n")
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        # rocprofv3 reports both in KiB. MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes for
        # wide coalesced streams -- double it; narrow accesses and WRITE_SIZE are uncalibrated. This kernel's traffic is register-spill
        # dwords and scattered model-constant reads, so both figures are given and the doubled one is the (upper) headline.
        fetch, write = out["FETCH_SIZE"] * 1024.0, out["WRITE_SIZE"] * 1024.0
        tj = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py --steps %d --warmup %d --no-extras --no-cpu-baseline (separate passes)" % (steps, bench["warmup"]),
              "kernel": bench["roofline"]["kernel"], "fetch_bytes_per_launch_raw": fetch, "fetch_bytes_per_launch_gfx950_corrected": 2 * fetch,
              "write_bytes_per_launch": write, "env_steps_per_launch": env_steps,
              "hbm_bytes_per_env_step": (2 * fetch + write) / env_steps, "hbm_bytes_per_env_step_raw_counters": (fetch + write) / env_steps,
              "algorithmic_bytes_per_env_step": bench["roofline"]["bytes_per_env_step"],
              "ratio_to_algorithmic": (2 * fetch + write) / env_steps / bench["roofline"]["bytes_per_env_step"]}
        with open(prefix + "_hbm_traffic.json", "w") as f:
            json.dump(tj, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
