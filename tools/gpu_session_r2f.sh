#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -k "merkle" > gpurun_out/pytest_gpu_r2f.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu_r2f.txt
timeout 900 python tools/time_paths.py > gpurun_out/time_paths_r2f.txt 2>&1; echo "time_paths rc=$?"
grep -E "leaves=|arity|build" gpurun_out/time_paths_r2f.txt
echo "--- HADES252_MERKLE_FUSE2=0"
HADES252_MERKLE_FUSE2=0 timeout 900 python tools/time_paths.py > gpurun_out/time_paths_r2f_nofuse.txt 2>&1
grep -E "leaves=|arity|build" gpurun_out/time_paths_r2f_nofuse.txt
