#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
./build_tools/ubench3 coop > gpurun_out/ubench3_coop_r2.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -k "coop or merkle or ragged or host or concurrent or bytes" > gpurun_out/pytest_gpu_r2c.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu_r2c.txt
timeout 900 python tools/time_paths.py > gpurun_out/time_paths_r2c.txt 2>&1; echo "time_paths rc=$?"
