#!/bin/bash
# Step time of one bench workload against the values of one environment switch of the library:
#   tools/env_sweep.sh WAVENET_HIP_RANK_SUB "0 32 64 128"     (on the GPU box, from the repository root)
set -u
VAR=$1
mkdir -p gpurun_out/sweep
for c in $2; do
  out=$(env $VAR=$c timeout -k 10 120 python bench.py --workload ${WORKLOAD:-2d512} --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline ${EXTRA:-}) || { echo "$VAR=$c FAILED"; exit 1; }
  echo "$VAR=$c $(echo "$out" | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' ')" | tee -a gpurun_out/sweep/env.log
done
