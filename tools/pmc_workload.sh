#!/bin/bash
# SQ / TCC counter passes over one bench workload (one counter group per run), on the GPU box:
#   tools/pmc_workload.sh <workload> <tag> [groups]     e.g. tools/pmc_workload.sh 1d1024 r02 "fetch write sq1 sq2"
# The per-dispatch tables land in gpurun_out/pmc/<tag>_<workload>_<group>.csv.
set -u
W=$1
TAG=$2
GROUPS_WANTED=${3:-"fetch write sq1 sq2"}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -o r -- python $ROOT/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-single-batch --no-exchange-probe --no-coach-probe > /dev/null 2> $OUT/${TAG}_${W}_$name.err || { echo "pass $name failed"; tail -5 $OUT/${TAG}_${W}_$name.err; exit 1; }
  cp $(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1) $OUT/${TAG}_${W}_$name.csv
  echo "pass $name done"
}
for grp in $GROUPS_WANTED; do
  case $grp in
    fetch) run fetch FETCH_SIZE || exit 1 ;;
    write) run write WRITE_SIZE || exit 1 ;;
    sq1) run sq1 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM || exit 1 ;;
    tccw) run tccw TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_REQ_sum || exit 1 ;;
    tccr) run tccr TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum || exit 1 ;;
    sq3) run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES || exit 1 ;;
    sq2) run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM || exit 1 ;;
  esac
done
