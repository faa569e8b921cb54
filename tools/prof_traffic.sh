#!/bin/bash
# PMC passes behind profiles/traffic.json: the default bench workload on both trees, closest-hit and any-hit,
# one counter group per run (rocprofv3 does not split counters over passes; no trace domain is mixed in).
# usage (on the GPU box): tools/prof_traffic.sh <outdir-under-gpurun_out> [builders: "sah reference"] [modes: "closest shadow"]
# then, back in the container:  python tools/make_traffic.py gpurun_out/<outdir>  ->  profiles/traffic.json + summaries
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
BUILDERS=${2:-"sah reference"}
MODES=${3:-"closest shadow"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE GRBM_GUI_ACTIVE"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum"
)
for b in $BUILDERS; do
 for m in $MODES; do
  D=$OUT/${b}_${m}
  mkdir -p $D
  i=0
  for p in "${PASSES[@]}"; do
   i=$((i+1))
   timeout -s KILL 240 rocprofv3 --pmc $p --output-format csv -d $D/pass$i -- python3 $R/bench.py --no-cpu-baseline --no-reference-tree --no-render --no-other-accels --no-strong-projection --no-two-streams --steps 5 --warmup 1 --builder $b --mode $m > $D/pass$i.log 2>&1 || { echo "$b $m pass $i failed"; tail -5 $D/pass$i.log; exit 1; }
   echo "$b $m pass $i done"
  done
  python3 $R/tools/pmc_summary.py $D > $D/summary.txt
  grep '^{' $D/pass1.log | tail -1 > $D/bench_line.json
  find $D -name "*counter_collection.csv" -delete
  find $D -name "*agent_info.csv" -delete
 done
done
echo all passes done
