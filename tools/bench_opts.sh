#!/bin/bash
# usage: tools/bench_opts.sh <log-prefix> "<bench args A>" "<bench args B>" ...
P=$1; shift
i=0
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $a > gpurun_out/${P}_$i.log 2>&1 || { echo "[$a] FAILED"; tail -3 gpurun_out/${P}_$i.log; exit 1; }
  python - "$a" gpurun_out/${P}_$i.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[2]) if x.startswith('{')][-1]
j=json.loads(l)
print('[%s]'%sys.argv[1], j['value'], 'Mrays/s kernel_ms', j['roofline']['kernel_ms'], 'ms/step', j['ms_per_step'])
PY
done
