#!/bin/bash
# The full measurement on the GPU box: the default bench (every workload, with the CPU baseline), every workload
# again under rocprofv3 --kernel-trace --stats, and the PMC passes (one counter group per run: HBM-side fetch and
# write sizes for every workload, instruction and cycle counters for the two single-kernel ones).  Results land in
# gpurun_out/measure, gpurun_out/trace and gpurun_out/pmc; tools/collect_profiles.py copies the summaries into profiles/.
#   tools/measure_round.sh <tag> [bench|trace|pmc|all]        (from the repository root, on the GPU box; one part per call
#   keeps a call inside a 20-minute limit)
set -u
TAG=${1:-r03}
PART=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/measure
mkdir -p $OUT
if [ $PART = bench ] || [ $PART = all ]; then
timeout -k 10 600 python $ROOT/bench.py > $OUT/${TAG}_bench_all.jsonl 2> $OUT/${TAG}_bench_all.err || { echo "bench failed"; tail -5 $OUT/${TAG}_bench_all.err; exit 1; }
echo "bench done"
fi
if [ $PART = trace ] || [ $PART = all ]; then
for w in 2d64 2d64n16 1d1024 2d512 2d512n16 1d1m 1d1mb64; do
  tools/trace_workload.sh $w $TAG || exit 1
done
fi
if [ $PART = pmc ] || [ $PART = all ]; then
for w in 2d64 1d1024; do
  tools/pmc_workload.sh $w $TAG "fetch write sq1 sq2" || exit 1
done
for w in 2d64n16 2d512 2d512n16 1d1m 1d1mb64; do
  tools/pmc_workload.sh $w $TAG "fetch write" || exit 1
done
fi
echo "measurement done ($PART)"
