#!/bin/bash
# rocprofv3 --kernel-trace --stats over one bench workload, on the GPU box:
#   tools/trace_workload.sh <workload> <tag>
# leaves gpurun_out/trace/<tag>_bench_<workload>_kernel_stats.csv and the bench line beside it.
set -u
W=$1
TAG=$2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$W
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$W -o r -- python $ROOT/bench.py --workload $W --no-cpu-baseline > $OUT/${TAG}_bench_${W}_traced.json 2> $OUT/${TAG}_${W}.err || { echo "trace of $W failed"; tail -5 $OUT/${TAG}_${W}.err; exit 1; }
cp $(find /tmp/trace_$W -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_${W}_kernel_stats.csv
echo "trace of $W done"
