#!/bin/bash
# The matrix of tools/host_path_torch_probe.py: one process per configuration.  usage: bash tools/host_path_torch_probe.sh > out.txt
cd "$(dirname "$0")/.."
P="python3 tools/host_path_torch_probe.py"
$P "no torch, system runtime" --no-torch
$P "torch first (bundled runtime)"
HSA_ENABLE_SDMA=0 $P "torch first, HSA_ENABLE_SDMA=0"
HSA_ENABLE_SDMA=1 $P "torch first, HSA_ENABLE_SDMA=1"
GPU_MAX_HW_QUEUES=2 $P "torch first, GPU_MAX_HW_QUEUES=2"
GPU_MAX_HW_QUEUES=8 $P "torch first, GPU_MAX_HW_QUEUES=8"
HSA_ENABLE_INTERRUPT=0 $P "torch first, HSA_ENABLE_INTERRUPT=0"
AMD_DIRECT_DISPATCH=0 $P "torch first, AMD_DIRECT_DISPATCH=0"
HIP_FORCE_DEV_KERNARG=1 $P "torch first, HIP_FORCE_DEV_KERNARG=1"
$P "system runtime bound first, torch after" --system-runtime-first
HSA_ENABLE_SDMA=0 $P "no torch, HSA_ENABLE_SDMA=0" --no-torch
$P "no torch, system runtime (again)" --no-torch
# chunk size of the page-locked pipeline (default n / 32 clamped to 2^16 .. 2^18 states = 10 .. 40 MiB): does the bundled
# runtime overlap smaller copies?  (the staging-thread path, which always moves 2^16-state chunks, is FASTER than the
# page-locked path under torch: 20.6 ms against 29.6)
for c in 16384 32768 65536 131072 262144; do
  HADES252_HOST_CHUNK=$c $P "torch first, HADES252_HOST_CHUNK=$c"
done
for c in 32768 65536 262144; do
  HADES252_HOST_CHUNK=$c $P "no torch, HADES252_HOST_CHUNK=$c" --no-torch
done
