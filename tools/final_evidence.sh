#!/bin/bash
# The round's evidence set in one call (on the GPU box): PMC passes behind traffic.json, counters of the binary-BVH walk, the
# default bench line, kernel-trace stats.   usage: tools/final_evidence.sh <tag>   ->   gpurun_out/<tag>_{traffic,pmc_bvh,stats}, <tag>_bench.json
set -u
T=$1
R=$GRAFT_REPO_ROOT
$R/tools/prof_traffic.sh ${T}_traffic > $R/gpurun_out/${T}_traffic.log 2>&1 || { tail -5 $R/gpurun_out/${T}_traffic.log; exit 1; }
echo traffic done
$R/tools/prof_pmc.sh ${T}_pmc_bvh --accel bvh --no-two-streams --steps 3 --warmup 1 > $R/gpurun_out/${T}_pmc_bvh.log 2>&1 || { tail -5 $R/gpurun_out/${T}_pmc_bvh.log; exit 1; }
find $R/gpurun_out/${T}_pmc_bvh -name "*counter_collection.csv" -delete; find $R/gpurun_out/${T}_pmc_bvh -name "*agent_info.csv" -delete
echo pmc bvh done
cd $R && timeout -k 10 400 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail -5 gpurun_out/${T}_bench.err; exit 1; }
echo bench done
$R/tools/prof_stats.sh ${T}_stats > $R/gpurun_out/${T}_stats.log 2>&1 || { tail -5 $R/gpurun_out/${T}_stats.log; exit 1; }
echo stats done
