#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r2g.txt 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/pytest_gpu_r2g.txt
timeout 900 python tools/time_paths.py > gpurun_out/time_paths_r2g.txt 2>&1; echo "time_paths rc=$?"
grep -E "witness|trace" gpurun_out/time_paths_r2g.txt
