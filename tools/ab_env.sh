#!/bin/bash
# usage: ab_env.sh "ENV1=a ENV2=b" "ENV1=c" ... ; the quick headline bench once per environment setting, two rounds
F="--no-cpu-baseline --no-reference-tree --no-render --no-other-accels --no-strong-projection --no-two-streams --steps 10 --warmup 2 $BENCH_EXTRA"
for rep in 1 2; do
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('[$e]', 'rep$rep', d['value'], 'Mrays/s', 'kernel_ms', r['kernel_ms'], 'chk', d['config']['hits_checksum'], 'nodes', r['algorithmic']['nodes_per_ray'], 'bin_ms', r['binning_ms'], 'anyhit', d.get('any_hit', {}).get('value'))
" || exit 1
done
done
