#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + separate PMC passes around bench.py.
# usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/pmc_sq.log
# the latency kernels on one state + the 2^16-leaf tree (kernel durations only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_latency -- python3 tools/lat_one.py > /dev/null 2> $OUT/trace_latency.log
find $OUT/trace_latency -name "*kernel_stats.csv" | head -1 | xargs cat > $OUT/latency_kernel_stats.csv
# ... and their instruction counts (VALU instructions per wave: the latency of a lone wave IS its instruction count)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_latency -- python3 tools/lat_one.py > /dev/null 2> $OUT/pmc_latency.log
python3 - "$OUT" <<'PY' > $OUT/latency_kernel_instructions.txt 2>&1
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "pmc_latency", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0]
        if not k.startswith(("k_", "void k_", "hades::k_")):
            continue
        d = acc.setdefault((k, r.get("Grid_Size", "")), {})
        d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("kernel, grid (threads): per-dispatch averages of SQ_WAVES, and VALU / SALU / LDS instructions PER WAVE")
for (k, g), d in sorted(acc.items()):
    w = sum(d.get("SQ_WAVES", [0])) / max(len(d.get("SQ_WAVES", [1])), 1)
    def per_wave(name):
        v = d.get(name)
        return (sum(v) / len(v)) / w if v and w else float("nan")
    print("%-46s grid %-8s n=%-4d waves %6.0f  VALU/wave %9.0f  SALU/wave %8.0f  LDS/wave %7.0f"
          % (k[:46], g, len(d.get("SQ_WAVES", [])), w, per_wave("SQ_INSTS_VALU"), per_wave("SQ_INSTS_SALU"), per_wave("SQ_INSTS_LDS")))
PY
cat $OUT/latency_kernel_instructions.txt
# the kernels behind bench.py's secondary rooflines, at bench.py's sizes: timings plain, HBM bytes in two more passes
python3 tools/secondary_kernels.py > $OUT/secondary_kernels.json 2> $OUT/secondary_kernels.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_sec_fetch -- python3 tools/secondary_kernels.py > /dev/null 2> $OUT/pmc_sec_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_sec_write -- python3 tools/secondary_kernels.py > /dev/null 2> $OUT/pmc_sec_write.log
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# wire-format kernels beyond the Infinity Cache, with HBM byte counters
python3 tools/wire_bw.py > $OUT/wire_bw.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_wire_fetch -- python3 tools/wire_bw.py > /dev/null 2> $OUT/pmc_wire_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_wire_write -- python3 tools/wire_bw.py > /dev/null 2> $OUT/pmc_wire_write.log
python3 - "$OUT" <<'PY' >> $OUT/wire_bw.txt 2>&1
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for d in ("pmc_wire_fetch", "pmc_wire_write"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "k_wire" in k or "bytes" in k:
                acc.setdefault((k.split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    avg = sum(v) / len(v)
    b = avg * 1024 * (2 if c == "FETCH_SIZE" else 1)
    print("%-14s %-10s n=%d  avg counter %.6g  -> %.4g bytes per launch%s (algorithmic: 2.147e9)"
          % (k, c, len(v), avg, b, " (x2 gfx950 wide-read correction)" if c == "FETCH_SIZE" else ""))
PY
cat $OUT/wire_bw.txt
# the VALU ceiling microbenchmark under the same counters (program directly after --)
if [ -x build_tools/ubench3 ]; then
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_ubench -- ./build_tools/ubench3 rates > $OUT/ubench3_rates_under_pmc.txt 2> $OUT/pmc_ubench.log
  python3 - "$OUT" <<'PY' > $OUT/ubench3_pmc_summary.txt 2>&1
import csv, glob, os, sys
out = sys.argv[1]
rows = {}
order = []
for f in sorted(glob.glob(os.path.join(out, "pmc_ubench", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        key = (r.get("Dispatch_Id"), r.get("Kernel_Name", "")[:40], r.get("Grid_Size"))
        if key not in rows:
            rows[key] = {}
            order.append(key)
        rows[key][r["Counter_Name"]] = rows[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("dispatch kernel grid | SQ_WAVES SQ_INSTS_VALU instr/wave SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU  (ubench3 rates, in launch order)")
for k in order:
    c = rows[k]
    w = c.get("SQ_WAVES", 0)
    print("%6s %-40s %8s | %8.0f %14.0f %10.1f %14.0f %14.0f" % (k[0], k[1], k[2], w, c.get("SQ_INSTS_VALU", 0),
          c.get("SQ_INSTS_VALU", 0) / w if w else 0, c.get("SQ_BUSY_CYCLES", 0), c.get("SQ_ACTIVE_INST_VALU", 0)))
PY
  cat $OUT/ubench3_pmc_summary.txt | head -60
fi
