#!/bin/bash
# Copies what one tools/gpu_session.sh run left under gpurun_out/ into profiles/<tag>/ (tracked) and installs the stamped
# counter record as profiles/hbm_traffic.json.  Curated notes in profiles/<tag>/ written by hand from probe outputs are not
# touched.  Files a session did not produce (the optional stand-alone probes) are skipped.
#   bash tools/collect_profiles.sh r5
set -u
TAG=${1:-r5}
cd "$(dirname "$0")/.."
G=gpurun_out
P=profiles/$TAG
mkdir -p $P
# the counter record: stamped with the measured commit (refused unless HEAD's kernel sources hash to the record's key)
python3 tools/stamp_profile.py $G/prof_$TAG/hbm_traffic.json $P || exit 1
c() { [ -e "$1" ] && cp "$1" "$2" || echo "skipped (not produced): $1"; }
c $G/bench_$TAG.json $P/bench_N1.json
c $G/bench_${TAG}_after_profile.json $P/bench_N1_after_profile.json
c $G/prof_$TAG/bench_trace.json $P/bench_N1_under_rocprof.json
c $G/prof_$TAG/pmc_summary.json $P/pmc_summary.json
c $G/prof_$TAG/summary.txt $P/rocprofv3_bench_2p26_summary.txt
c $G/prof_$TAG/latency_kernel_stats.csv $P/latency_kernel_stats.csv
c $G/prof_$TAG/latency_kernel_instructions.txt $P/latency_kernel_instructions.txt
c $G/prof_$TAG/secondary_kernels.json $P/secondary_kernels_timings.json
find $G/prof_$TAG/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/kernel_stats.csv
c $G/prof_$TAG/wire_bw.txt $P/wire_bw_last_session.txt
c $G/time_paths_$TAG.txt $P/time_paths.txt
c $G/host_callers_$TAG.txt $P/host_callers.txt
c $G/host_path_$TAG.txt $P/host_path_native.txt
c $G/host_path_torch_probe_$TAG.txt $P/host_path_torch_probe.txt
c $G/bench_rehearsal_2ranks.json $P/bench_rehearsal_2ranks_one_device.json
c $G/bench_rehearsal_8ranks.json $P/bench_rehearsal_8ranks_one_device.json
# round 6: BASELINE configs[4] as a strong-scaling measurement (tools/gpu_session_r6a.sh)
c $G/r6a/bench_strong_2p30_N1.json $P/bench_strong_2p30_N1.json
c $G/r6a/bench_strong_rehearsal_2ranks.json $P/bench_strong_rehearsal_2ranks_one_device.json
for f in lanes_proto pcie_probe residency ubench3 dfma_proto wire_proto copy_proto pin_probe; do c $G/${f}_$TAG.txt $P/$f.txt; done
tail -6 $G/pytest_gpu_$TAG.txt > $P/pytest_gpu_summary.txt
ls -la $P
