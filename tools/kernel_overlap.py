#!/usr/bin/env python3
"""How much of the non-engine GPU work of a run overlaps with engine launches?  Reads a rocprofv3 --kernel-trace CSV (Kernel_Name, Start_Timestamp, End_Timestamp).

For the pipelined config-5 loop (agent.BatchedGraspAgent(pipeline_groups=2)): the CNN / renderer / optimiser kernels of scene group g + 1 should run INSIDE group g's
grasp launch (ur5m_run_kernel). Prints one JSON line: per kernel family the busy time (union of its intervals) and the part of it covered by an engine launch, plus
the wall time of the trace and the share of it during which an engine kernel was running.
    python tools/kernel_overlap.py <kernel_trace.csv> [engine kernel name fragment = run_kernel]"""
import csv, json, sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def covered(iv, cover):
    """length of the part of the (merged) intervals iv that lies inside the (merged) intervals cover"""
    t, j = 0, 0
    for a, b in iv:
        while j < len(cover) and cover[j][1] <= a:
            j += 1
        k = j
        while k < len(cover) and cover[k][0] < b:
            t += max(0, min(b, cover[k][1]) - max(a, cover[k][0]))
            k += 1
    return t


def family(name):
    n = name.lower()
    if "run_kernel" in n: return "engine (grasp / settle launches)"
    if "render" in n: return "renderer"
    if "conv" in n or "gemm" in n or "cijk" in n or "miopen" in n or "winograd" in n or "sp3" in n or "batchnorm" in n or "bn_" in n: return "CNN: convolutions / GEMMs / batch norm"
    return "other torch kernels (elementwise, reductions, optimiser, indexing)"


rows = list(csv.DictReader(open(sys.argv[1], newline="")))
frag = sys.argv[2] if len(sys.argv) > 2 else "run_kernel"
iv = {}
for r in rows:
    iv.setdefault(family(r["Kernel_Name"]), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
eng = union(iv.get("engine (grasp / settle launches)", []))
t0 = min(a for v in iv.values() for a, _ in v); t1 = max(b for v in iv.values() for _, b in v)
out = {"trace_wall_s": (t1 - t0) * 1e-9, "engine_busy_s": sum(b - a for a, b in eng) * 1e-9, "engine_share_of_wall": sum(b - a for a, b in eng) / (t1 - t0), "families": {}}
for fam, v in sorted(iv.items()):
    if fam.startswith("engine"): continue
    u = union(v)
    busy = sum(b - a for a, b in u)
    out["families"][fam] = {"kernels": len(v), "busy_s": busy * 1e-9, "inside_an_engine_launch_s": covered(u, eng) * 1e-9, "inside_share": covered(u, eng) / busy if busy else None}
allother = union([x for fam, v in iv.items() if not fam.startswith("engine") for x in v])
busy = sum(b - a for a, b in allother)
out["all_non_engine"] = {"busy_s": busy * 1e-9, "inside_an_engine_launch_s": covered(allother, eng) * 1e-9, "inside_share": covered(allother, eng) / busy if busy else None}
print(json.dumps(out))


def timeline(rows, out=sys.stderr, gap_ns=2_000_000):
    """compact timeline on stderr: engine launches one per line, the other kernels as bursts (runs separated by < gap_ns) per queue"""
    ev = []
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    qcol = "Queue_Id" if "Queue_Id" in rows[0] else None
    bursts = {}
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        a, b, q = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, (r[qcol] if qcol else "?")
        if "run_kernel" in r["Kernel_Name"]:
            ev.append((a, "ENGINE  q%-3s %9.1f .. %9.1f ms  (%7.1f ms)" % (q, a / 1e6, b / 1e6, (b - a) / 1e6)))
            continue
        cur = bursts.get(q)
        if cur and a - cur[1] < gap_ns:
            cur[1] = max(cur[1], b); cur[2] += 1; cur[3] += b - a
        else:
            if cur: ev.append((cur[0], "burst   q%-3s %9.1f .. %9.1f ms  %5d kernels, busy %7.1f ms" % (q, cur[0] / 1e6, cur[1] / 1e6, cur[2], cur[3] / 1e6)))
            bursts[q] = [a, b, 1, b - a]
    for q, cur in bursts.items():
        ev.append((cur[0], "burst   q%-3s %9.1f .. %9.1f ms  %5d kernels, busy %7.1f ms" % (q, cur[0] / 1e6, cur[1] / 1e6, cur[2], cur[3] / 1e6)))
    for _, line in sorted(ev):
        print(line, file=out)


if len(sys.argv) > 3 and sys.argv[3] == "timeline":
    timeline(rows)
