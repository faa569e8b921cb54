#!/bin/bash
# Kernel-trace stats of the wavefront renderer on the 10.5M-triangle courtyard at 4K.
# usage (on the GPU box): tools/prof_render.sh <outdir-under-gpurun_out> [bench_render args]
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/bench_render.py "$@" > $OUT/stats.log 2>&1 || { echo "stats failed"; tail -5 $OUT/stats.log; exit 1; }
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name "*kernel_trace.csv" -delete
tail -1 $OUT/stats.log
cut -d, -f1-4,7 $OUT/kernel_stats.csv | head -20
