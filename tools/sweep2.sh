#!/bin/bash
set -e
WL=${1:-c3}
for bpc in 2 4 5 8; do for un in 4 8; do for nt in 0 1; do
  echo "== blocks_per_cu=$bpc unroll=$un nt=$nt"
  VFSX_BLOCKS_PER_CU=$bpc VFSX_SCAN_UNROLL=$un VFSX_SCAN_NT=$nt timeout -k 10 120 python -u bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['avg_kernel_ms'])"
done; done; done
