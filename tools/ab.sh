#!/bin/bash
# A/B timing of library variants (development): tools/ab.sh <n> variant.so...
N=${1:-16777216}; shift
for round in 1 2; do
  for v in "$@"; do
    printf "%-28s " $(basename $v)
    HADES252_LIB=$PWD/$v python tools/time_kernel.py $N 2 2>&1 | tail -1
  done
done
