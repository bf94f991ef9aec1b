#!/bin/bash
# The full measurement on the GPU box: default bench (with the CPU baseline), the same command under
# rocprofv3 --kernel-trace --stats, and the PMC passes (one counter group per run).  Results land in
# gpurun_out/measure/; tools/collect_profiles.py copies the summaries into profiles/.
#   tools/measure_round.sh        (from the repository root, on the GPU box)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/measure
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $ROOT/bench.py > $OUT/bench_2d64_n4.json 2> $OUT/bench_2d64_n4.err || { echo "bench failed"; exit 1; }
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o r -- python $ROOT/bench.py --no-cpu-baseline > $OUT/bench_2d64_n4_traced.json 2> $OUT/trace.err || { echo "trace failed"; exit 1; }
cp $(find /tmp/prof_trace -name "*kernel_stats.csv" | head -1) $OUT/bench_2d64_n4_kernel_stats.csv
echo "trace done"
i=0
for counters in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d /tmp/prof_pmc$i -o r -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; exit 1; }
  cp $(find /tmp/prof_pmc$i -name "*counter_collection.csv" | head -1) $OUT/pmc$i.csv
  echo "pmc pass $i done"
done
for w in 1d1024 2d512 1d1m; do
  timeout -k 10 300 python $ROOT/bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { echo "bench $w failed"; exit 1; }
  echo "bench $w done"
done
