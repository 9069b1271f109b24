#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
./build_tools/dfma_proto > gpurun_out/dfma_proto_r2.txt 2>&1; echo "dfma rc=$?"; cat gpurun_out/dfma_proto_r2.txt
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x -k "coop or merkle or ragged or host or concurrent or bytes or kats or edge" > gpurun_out/pytest_gpu_r2e.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu_r2e.txt
timeout 900 python tools/time_paths.py > gpurun_out/time_paths_r2e.txt 2>&1; echo "time_paths rc=$?"
grep -E "coop|leaves=|host call|arity" gpurun_out/time_paths_r2e.txt
