#!/bin/bash
# usage: tools/bench_variants.sh <log-prefix> name1 name2 ...   (libs in variants/libcrt_<name>.so; "main" = in-tree lib)
P=$1; shift
for n in "$@"; do
  if [ "$n" = main ]; then unset CRT_GPU_LIB; else export CRT_GPU_LIB=$GRAFT_REPO_ROOT/variants/libcrt_$n.so; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${P}_$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/${P}_$n.log; exit 1; }
  python - "$n" gpurun_out/${P}_$n.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith('{')][-1]
j=json.loads(l)
print(sys.argv[1], j["value"], "chk", j["config"].get("hits_checksum"), 'Mrays/s kernel_ms', j['roofline']['kernel_ms'], 'nodes', j['roofline']['nodes_per_ray'], 'tris', j['roofline']['tris_per_ray'])
PY
done
