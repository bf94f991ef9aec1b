#!/bin/bash
# Runs the bench with each library variant under wavenet_amd/lib/exp and prints kernel time.
#   [WORKLOAD=2d64] [TAPS="4 8"] tools/ab_bench.sh
cd "$(dirname "$0")/.."
WORKLOAD=${WORKLOAD:-2d64}
TAPS=${TAPS:-"4 8"}
for so in wavenet_amd/lib/exp/*.so; do
  for taps in $TAPS; do
    out=$(WAVENET_HIP_LIBRARY=$PWD/$so timeout -k 10 120 python bench.py --workload $WORKLOAD --steps 20 --warmup 3 --no-cpu-baseline --taps $taps) || { echo "$so taps $taps FAILED"; exit 1; }
    echo "$(basename $so) taps=$taps $(echo "$out" | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' ')"
  done
done
