#!/bin/bash
# round-2 GPU session A: parity tier, ubench3, residency sweep, paths, bench
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r2a.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu_r2a.txt
tail -5 gpurun_out/pytest_gpu_r2a.txt
timeout 600 ./build_tools/ubench3 > gpurun_out/ubench3_r2.txt 2>&1; echo "ubench3 rc=$?"
timeout 300 ./build_tools/residency > gpurun_out/residency_r2.txt 2>&1; echo "residency rc=$?"
timeout 900 python tools/time_paths.py > gpurun_out/time_paths_r2a.txt 2>&1; echo "time_paths rc=$?"
timeout 600 python bench.py > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/bench_r2a.json
