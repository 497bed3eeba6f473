#!/bin/bash
# a bench workload at several result limits: $1 workload, rest = the k values (GPU box)
w=$1; shift
for k in "$@"; do
timeout -k 10 200 python3 bench.py --workload $w --k $k --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$w k=$k', 'value', d['value'], 'kernel_ms', r['avg_kernel_ms'], 'lat', (d.get('query_latency_ms') or {}).get('median'))" || exit 1
done
