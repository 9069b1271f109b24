#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/profile_round.sh r2 > gpurun_out/profile_round_r2.log 2>&1; echo "profile rc=$?"
tail -40 gpurun_out/profile_round_r2.log
