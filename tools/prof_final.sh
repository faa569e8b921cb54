#!/bin/bash
# Round profile set: kernel-trace stats of the default bench command, FETCH_SIZE calibration, PMC passes.
# usage (on the GPU box): tools/prof_final.sh <outdir-under-gpurun_out>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-reference-tree > $OUT/stats.log 2>&1 || { echo "stats failed"; tail -5 $OUT/stats.log; exit 1; }
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name "*kernel_trace.csv" -delete
# the same command without the render leg: the path tracer launches the same walk kernels on smaller queues, which
# would pull their average duration away from the headline launch's; this pass is the one roofline.kernel_ms must match
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_walk -- python3 $R/bench.py --no-cpu-baseline --no-reference-tree --no-render > $OUT/stats_walk.log 2>&1 || { echo "stats_walk failed"; tail -5 $OUT/stats_walk.log; exit 1; }
find $OUT/stats_walk -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_walk.csv \;
find $OUT/stats_walk -name "*kernel_trace.csv" -delete
timeout -s KILL 120 $R/tools/calib_fetch > $OUT/calib_plain.log 2>&1 || { echo "calib failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- $R/tools/calib_fetch > $OUT/calib_pmc.log 2>&1 || { echo "calib pmc failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d $OUT/calib2 -- $R/tools/calib_fetch > $OUT/calib_pmc2.log 2>&1 || { echo "calib pmc2 failed"; exit 1; }
cat $OUT/calib_plain.log
python3 - <<PY
import csv, glob
for d in ("calib", "calib2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            print(d, row["Kernel_Name"][:30], row["Counter_Name"], row["Counter_Value"])
PY
