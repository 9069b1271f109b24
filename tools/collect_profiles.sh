#!/bin/bash
# Copies what one tools/gpu_session.sh run left under gpurun_out/ into profiles/<tag>/ (tracked).  The curated notes
# there (host_path.txt, lds_vs_sgpr_constants.txt, wire_bw.txt, forest_streams_probe.txt) are written by hand from the
# probes' outputs and are not touched.
#   bash tools/collect_profiles.sh r3
set -eu
TAG=${1:-r3}
cd "$(dirname "$0")/.."
G=gpurun_out
P=profiles/$TAG
mkdir -p $P
cp $G/bench_$TAG.json $P/bench_N1.json
cp $G/prof_$TAG/bench_trace.json $P/bench_N1_under_rocprof.json
# the counter record: stamped with the measured commit (refused unless HEAD's kernel sources hash to the record's key)
python3 tools/stamp_profile.py $G/prof_$TAG/hbm_traffic.json $P
cp $G/prof_$TAG/pmc_summary.json $P/pmc_summary.json
cp $G/prof_$TAG/summary.txt $P/rocprofv3_bench_2p26_summary.txt
cp $G/prof_$TAG/latency_kernel_stats.csv $P/latency_kernel_stats.csv
cp $G/prof_$TAG/latency_kernel_instructions.txt $P/latency_kernel_instructions.txt
find $G/prof_$TAG/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/kernel_stats.csv
cp $G/prof_$TAG/wire_bw.txt $P/wire_bw_last_session.txt
cp $G/time_paths_$TAG.txt $P/time_paths.txt
cp $G/lanes_proto_$TAG.txt $P/lanes_proto.txt
cp $G/pcie_probe_$TAG.txt $P/pcie_probe.txt
cp $G/residency_$TAG.txt $P/residency.txt
cp $G/host_callers_$TAG.txt $P/host_callers.txt
tail -6 $G/pytest_gpu_$TAG.txt > $P/pytest_gpu_summary.txt
ls -la $P
