#!/bin/bash
# usage: tools/bench_variants_shadow.sh name1 name2 ...   (libs in variants/libcrt_<name>.so; "main" = in-tree lib)
for n in "$@"; do
  if [ "$n" = main ]; then unset CRT_GPU_LIB; else export CRT_GPU_LIB=$GRAFT_REPO_ROOT/variants/libcrt_$n.so; fi
  timeout -k 10 300 python bench.py --mode shadow --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/sh_$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/sh_$n.log; exit 1; }
  python - "$n" gpurun_out/sh_$n.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith('{')][-1]
j=json.loads(l)
print(sys.argv[1], j['value'], "chk", j["config"].get("hits_checksum"), 'kernel_ms', j['roofline']['kernel_ms'])
PY
done
