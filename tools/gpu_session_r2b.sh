#!/bin/bash
# round-2 GPU session B: full parity tier, paths, bench
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r2b.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu_r2b.txt
tail -15 gpurun_out/pytest_gpu_r2b.txt
timeout 900 python tools/time_paths.py > gpurun_out/time_paths_r2b.txt 2>&1; echo "time_paths rc=$?"
timeout 600 python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; echo "bench rc=$?"
