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


def secondary_kernels(build):
    """HBM bytes of the kernels behind bench.py's `secondary` rooflines (tools/secondary_kernels.py under --pmc FETCH_SIZE
    and --pmc WRITE_SIZE, separate passes: pmc_sec_fetch / pmc_sec_write), per launch -- per TREE for the Merkle build
    (the sum over all of a tree's launches) -- corrected like the headline kernel's.  Keyed by build.device_source_hash()."""
    def rows(d):
        for f in find(d + "/**/*counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    yield r.get("Kernel_Name", ""), r.get("Counter_Name", ""), float(r.get("Counter_Value", 0))
    try:
        counts = json.loads(open(os.path.join(out, "secondary_kernels.json")).read().strip().splitlines()[-1])
    except Exception:
        return None
    trees = counts["merkle_2p24"]["trees_built"]
    groups = {"wire_to_bytes": lambda k: "k_wire<0>" in k, "wire_from_bytes": lambda k: "k_wire<1>" in k,
              "witness": lambda k: "k_perm_witness" in k, "trace": lambda k: "k_perm_trace_fast" in k,
              "trace_scaled": lambda k: "k_perm_trace_scaled" in k,
              "merkle_2p24_tree": lambda k: "merkle" in k}
    acc = {g: {"FETCH_SIZE": [], "WRITE_SIZE": []} for g in groups}
    for d in ("pmc_sec_fetch", "pmc_sec_write"):
        for k, c, v in rows(d):
            for g, pred in groups.items():
                if pred(k) and c in acc[g]:
                    acc[g][c].append(v)
    res = {"device_source_hash": build.device_source_hash(),
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes around `python3 "
                     "tools/secondary_kernels.py` (bench.py's own record functions at bench.py's sizes); bytes = counter*1024, "
                     "FETCH_SIZE doubled (gfx950 wide-read correction); per launch, the Merkle entry per tree build (sum "
                     "over the tree's launches)"}
    for g, cs in acc.items():
        if not cs["FETCH_SIZE"] or not cs["WRITE_SIZE"]:
            continue
        if g == "merkle_2p24_tree":
            if len(cs["FETCH_SIZE"]) % trees or len(cs["WRITE_SIZE"]) % trees:     # the counted builds must tile the dispatches
                print("secondary merkle_2p24_tree: %d / %d dispatches do not divide into %d trees -- record dropped"
                      % (len(cs["FETCH_SIZE"]), len(cs["WRITE_SIZE"]), trees))
                continue
            rd = sum(cs["FETCH_SIZE"]) / trees * 1024 * 2
            wr = sum(cs["WRITE_SIZE"]) / trees * 1024
            n = trees
        else:
            rd = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2
            wr = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
            n = len(cs["FETCH_SIZE"])
        res[g] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr, "launches_or_trees_averaged": n}
        if g == "merkle_2p24_tree":
            res[g]["launches_per_tree"] = len(cs["FETCH_SIZE"]) // trees
        print("secondary %-18s read %.5g B  write %.5g B  total %.5g B  (n=%d)" % (g, rd, wr, rd + wr, n))
    return res


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
# bench.py replays the per-launch HBM traffic of k_perm_fast from profiles/hbm_traffic.json -- only when the
# kernel source hash recorded here matches the library it runs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from hades252_amd import build as _hb
    for k, cs in summary.items():
        if "k_perm_fast" in k and "hbm_bytes_per_launch" in cs:
            perms = None
            try:
                perms = json.loads(open(os.path.join(out, "bench_fetch.json")).read().strip().splitlines()[-1])["config"]["perms_per_gpu"]
            except Exception:
                pass
            rec = {"kernel": "k_perm_fast", "perms_per_launch": perms, "kernel_source_hash": _hb.perm_fast_hash(),
                   "hbm_bytes_per_launch": cs["hbm_bytes_per_launch"], "hbm_read_bytes": cs["hbm_read_bytes"],
                   "hbm_write_bytes": cs["hbm_write_bytes"],
                   "valu_instructions_per_wave": (cs.get("SQ_INSTS_VALU", 0) / cs["SQ_WAVES"]) if cs.get("SQ_WAVES") else None,
                   "algorithmic_bytes_per_launch": 320 * perms if perms else None,
                   "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes around `python3 bench.py "
                             "--steps 5 --warmup 1 --no-cpu-baseline` (tools/profile_round.sh); bytes = counter*1024, "
                             "FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section)"}
            sec = secondary_kernels(_hb)
            if sec:
                rec["secondary_kernels"] = sec
            json.dump(rec, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
            print("\nwrote hbm_traffic.json:", rec)
except Exception as e:                     # pragma: no cover
    print("hbm_traffic.json not written:", e)
for f in ("bench_trace.json", "bench_fetch.json"):
    p = os.path.join(out, f)
    if os.path.exists(p):
        print("\n== %s ==" % f)
        print(open(p).read().strip()[-1500:])
