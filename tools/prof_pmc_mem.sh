#!/bin/bash
# memory-side PMC passes only: tools/prof_pmc_mem.sh <outdir> [bench args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=("TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum")
i=0
for p in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $p --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | grep -A12 "void k_bvh4"
