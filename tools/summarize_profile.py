#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into a small summary."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows:
        print("%-60s calls %6s  total %14s ns  avg %14s ns  pct %6s" % (
            r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))

kt = find("trace/**/*kernel_trace.csv")
durs = {}
regs = {}
for f in kt:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            durs.setdefault(name, []).append(d)
            regs[name] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                          r.get("Scratch_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
print("\n== per-kernel dispatch durations (kernel trace) ==")
for name, d in durs.items():
    d2 = sorted(d)
    print("%-60s n=%4d  avg %.3f ms  median %.3f ms  min %.3f ms   vgpr/agpr/sgpr/lds/scratch/wg/grid=%s" % (
        name[:60], len(d), sum(d) / len(d) / 1e6, d2[len(d2) // 2] / 1e6, d2[0] / 1e6, regs[name]))


def pmc(dirname):
    acc = {}
    for f in find(dirname + "/**/*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                key = (r.get("Kernel_Name", ""), r.get("Counter_Name", ""))
                acc.setdefault(key, []).append(float(r.get("Counter_Value", 0)))
    return acc


summary = {}
print("\n== PMC (separate passes) ==")
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for (k, c), vals in sorted(pmc(d).items()):
        if "perm" not in k and "merkle" not in k:
            continue
        avg = sum(vals) / len(vals)
        print("%-40s %-22s n=%3d avg %.6g" % (k[:40], c, len(vals), avg))
        summary.setdefault(k, {})[c] = avg

# HBM traffic per launch, corrected as MI355X_MICROARCH.md prescribes:
#   FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B (hbm_bytes = counter * 1024);
#   on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced read: double it.
for k, cs in summary.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        rd = cs["FETCH_SIZE"] * 1024 * 2
        wr = cs["WRITE_SIZE"] * 1024
        print("\n%s: HBM read %.4g B (FETCH_SIZE x1024 x2 gfx950 correction), write %.4g B, total %.4g B per launch"
              % (k[:50], rd, wr, rd + wr))
        summary[k]["hbm_read_bytes"] = rd
        summary[k]["hbm_write_bytes"] = wr
        summary[k]["hbm_bytes_per_launch"] = rd + wr
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for f in ("bench_trace.json", "bench_fetch.json"):
    p = os.path.join(out, f)
    if os.path.exists(p):
        print("\n== %s ==" % f)
        print(open(p).read().strip()[-1500:])
