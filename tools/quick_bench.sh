#!/bin/bash
# One line per workload with the current library: examples/s and device time per step.
#   tools/quick_bench.sh "2d512 1d1024"      (on the GPU box, from the repository root)
set -u
mkdir -p gpurun_out
WL=${1:-"2d64 2d64n16 1d1024 2d512 1d1m"}
for w in $WL; do
  out=$(timeout -k 10 200 python bench.py --workload $w --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${EXTRA:-}) || { echo "$w FAILED"; exit 1; }
  echo "$w $(echo "$out" | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' ')" | tee -a gpurun_out/quick.log
done
