#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command (with and without the render leg) and of the other accelerators' lines.
# usage (on the GPU box): tools/prof_stats.sh <outdir-under-gpurun_out>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; exit 1; }
  find $OUT/$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  grep '^{' $OUT/$name.log | tail -1 > $OUT/bench_line_$name.json
  rm -rf $OUT/$name
  echo "$name done"
}
run full
run walk_with_reference_tree --no-render --no-other-accels --no-two-streams --no-strong-projection
run walk --no-render --no-reference-tree --no-other-accels --no-two-streams --no-strong-projection   # the headline launches alone: this average is the one roofline.kernel_ms must match
run bvh --no-render --accel bvh --no-two-streams
run grid --no-render --accel grid --workload courtyard-1M --no-two-streams
