#!/bin/bash
# repeat a bench workload: $1 workload, $2 repetitions (GPU box)
for i in $(seq 1 ${2:-3}); do
timeout -k 10 200 python3 bench.py --workload $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$1', 'value', d['value'], 'kernel_ms', r['avg_kernel_ms'], 'frac', r['frac'], 'lat', (d.get('query_latency_ms') or {}).get('median'))"
done
