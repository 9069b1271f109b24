#!/bin/bash
# One GPU-box session: the whole GPU test tier, the timing table, the headline bench and the rocprofv3 round
# (kernel trace + separate PMC passes on bench.py and on the kernels behind its secondary records).
#   gpurun --timeout 3600 -- 'bash tools/gpu_session.sh r5'            (everything)
#   gpurun --timeout 3600 -- 'bash tools/gpu_session.sh r5 probes'     (+ the stand-alone probes of tools/*.hip)
set -u
TAG=${1:-r5}
PROBES=${2:-}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu_$TAG.txt
tail -4 gpurun_out/pytest_gpu_$TAG.txt
if [ -n "$PROBES" ]; then
  timeout 600 ./build_tools/ubench3 > gpurun_out/ubench3_$TAG.txt 2>&1; echo "ubench3 rc=$?"
  timeout 120 ./build_tools/lanes_proto > gpurun_out/lanes_proto_$TAG.txt 2>&1; echo "lanes_proto rc=$?"
  timeout 120 ./build_tools/pcie_probe 640 > gpurun_out/pcie_probe_$TAG.txt 2>&1; echo "pcie_probe rc=$?"
  timeout 300 ./build_tools/residency > gpurun_out/residency_$TAG.txt 2>&1; echo "residency rc=$?"
  timeout 120 ./build_tools/dfma_proto > gpurun_out/dfma_proto_$TAG.txt 2>&1; echo "dfma rc=$?"
  timeout 120 ./build_tools/wire_proto 26 > gpurun_out/wire_proto_$TAG.txt 2>&1; echo "wire_proto rc=$?"
  timeout 120 ./build_tools/copy_proto > gpurun_out/copy_proto_$TAG.txt 2>&1; echo "copy_proto rc=$?"
  timeout 60 ./build_tools/pin_probe > gpurun_out/pin_probe_$TAG.txt 2>&1; echo "pin_probe rc=$?"
fi
timeout 900 python tools/time_paths.py > gpurun_out/time_paths_$TAG.txt 2>&1; echo "time_paths rc=$?"
timeout 300 ./build_tools/host_path_bench callers > gpurun_out/host_callers_$TAG.txt 2>&1; echo "host callers rc=$?"
timeout 300 ./build_tools/host_path_bench 20 22 24 > gpurun_out/host_path_$TAG.txt 2>&1; echo "host path rc=$?"
timeout 600 bash tools/host_path_torch_probe.sh > gpurun_out/host_path_torch_probe_$TAG.txt 2>&1; echo "torch probe rc=$?"
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
bash tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1; echo "profile rc=$?"
# the record the profile round wrote is what bench.py replays from the NEXT line on (on the box only: the stamped copy is
# installed in the repository by tools/collect_profiles.sh)
cp gpurun_out/prof_$TAG/hbm_traffic.json gpurun_out/hbm_traffic_unstamped_$TAG.json 2>/dev/null
timeout 600 python bench.py > gpurun_out/bench_${TAG}_after_profile.json 2>> gpurun_out/bench_$TAG.err; echo "bench2 rc=$?"
tail -c 600 gpurun_out/bench_${TAG}_after_profile.json
