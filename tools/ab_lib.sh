#!/bin/bash
# A/B of two library builds on bench workloads (alternating arms on one box): $1 = other library, rest = workloads
set -o pipefail
OTHER=$1; shift
mkdir -p gpurun_out/ab
for W in "$@"; do
for i in 1 2; do
  for arm in default other; do
    if [ $arm = other ]; then export VFSX_LIB_PATH=$PWD/$OTHER; else unset VFSX_LIB_PATH; fi
    timeout -k 10 300 python3 bench.py --workload $W --no-cpu-baseline > gpurun_out/ab/$W.$arm.$i.json 2> gpurun_out/ab/$W.$arm.$i.err || { tail -5 gpurun_out/ab/$W.$arm.$i.err; exit 1; }
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ab/$W.$arm.$i.json') if l.startswith('{')][-1])
print('$W', '$arm', $i, 'value', d['value'], 'kernel_ms', d['roofline']['avg_kernel_ms'], 'frac', d['roofline']['frac'], 'fallbacks', d.get('mfma_fallback_queries'))"
  done
done
done
