#!/bin/bash
# Runs the default bench with each library variant under wavenet_amd/lib/exp and prints kernel time.
cd "$(dirname "$0")/.."
for so in wavenet_amd/lib/exp/*.so; do
  for taps in 4 8; do
    out=$(WAVENET_HIP_LIBRARY=$PWD/$so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --taps $taps) || { echo "$so taps $taps FAILED"; exit 1; }
    echo "$(basename $so) taps=$taps $(echo "$out" | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' ')"
  done
done
