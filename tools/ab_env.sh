#!/bin/bash
# sweep of one environment variable over a bench workload on one box: $1 = workload, $2 = variable, rest = values ("-" = unset)
set -o pipefail
W=$1; V=$2; shift 2
mkdir -p gpurun_out/ab
for rep in 1 2; do
for val in "$@"; do
  if [ "$val" = "-" ]; then unset $V; else export $V=$val; fi
  timeout -k 10 300 python3 bench.py --workload $W --no-cpu-baseline > gpurun_out/ab/env.json 2> gpurun_out/ab/env.err || { tail -5 gpurun_out/ab/env.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ab/env.json') if l.startswith('{')][-1])
print('$V=$val', 'value', d['value'], 'kernel_ms', d['roofline']['avg_kernel_ms'], 'frac', d['roofline']['frac'], 'lat', d.get('query_latency_ms',{}).get('median'), 'fallbacks', d.get('mfma_fallback_queries'), 'of', d.get('mfma_queries'))"
done
done
