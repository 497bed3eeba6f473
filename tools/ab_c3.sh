#!/bin/bash
# A/B of two library builds on the C3 single-query scan (alternating arms on one box): $1 = other library
set -o pipefail
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for arm in default other; do
    if [ $arm = other ]; then export VFSX_LIB_PATH=$PWD/$1; else unset VFSX_LIB_PATH; fi
    timeout -k 10 200 python3 bench.py --workload ${2:-c3} --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/$arm.$i.json 2> gpurun_out/ab/$arm.$i.err || exit 1
    python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ab/$arm.$i.json') if l.startswith('{')][-1])
print('$arm', $i, d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['query_latency_ms']['median'])"
  done
done
