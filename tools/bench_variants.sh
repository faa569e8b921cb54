#!/bin/bash
# usage: tools/bench_variants.sh <log-prefix> name1 name2 ...   (libs in variants/libcrt_<name>.so; "main" = in-tree lib)
# headline batch only (no render, no reference-tree leg, no cpu baseline); BENCH_ARGS adds bench.py flags
P=$1; shift
mkdir -p gpurun_out
for n in "$@"; do
  if [ "$n" = main ]; then unset CRT_GPU_LIB; else export CRT_GPU_LIB=$GRAFT_REPO_ROOT/variants/libcrt_$n.so; fi
  timeout -k 10 ${VARIANT_TIMEOUT:-300} python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-render --no-reference-tree --no-other-accels ${BENCH_ARGS:-} > gpurun_out/${P}_$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/${P}_$n.log; exit 1; }
  python3 - "$n" gpurun_out/${P}_$n.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith('{')][-1]
j=json.loads(l)
a=j['roofline']['algorithmic']
print(sys.argv[1], j["value"], "chk", j["config"].get("hits_checksum"), 'kernel_ms', j['roofline']['kernel_ms'], 'nodes', a['nodes_per_ray'], 'tris', a['tris_per_ray'], 'anyhit', j.get('any_hit',{}).get('value'))
PY
done
