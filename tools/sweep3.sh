#!/bin/bash
# pipelined vs plain scan variants (development aid): $1 workload
WL=${1:-c3}
for pipe in 0 1; do for un in 4 8; do for bpc in 4 8; do
  echo "== pipe=$pipe unroll=$un blocks_per_cu=$bpc"
  VFSX_SCAN_PIPE=$pipe VFSX_BLOCKS_PER_CU=$bpc VFSX_SCAN_UNROLL=$un timeout -k 10 120 python -u bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['avg_kernel_ms'])"
done; done; done
