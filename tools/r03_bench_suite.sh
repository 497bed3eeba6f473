#!/bin/bash
# round-3 measurement suite (GPU box): every line's JSON lands under gpurun_out/r03/
set -o pipefail
mkdir -p gpurun_out/r03
run() { # name, timeout, args...
  local name=$1 t=$2; shift 2
  echo "== $name" >&2
  timeout -k 10 "$t" python3 bench.py "$@" > "gpurun_out/r03/$name.json" 2> "gpurun_out/r03/$name.err" || { echo "FAILED $name rc=$?" >&2; tail -5 "gpurun_out/r03/$name.err" >&2; return 1; }
  tail -c 600 "gpurun_out/r03/$name.json" >&2; echo >&2
}
case "${1:-a}" in
a)
run c2 300 --gpus 1 --steps 20 --warmup 5 &&
run c5 300 --workload c5 --steps 10 --warmup 2 --no-cpu-baseline &&
run c3 300 --workload c3 --steps 20 --warmup 5 &&
run c1 300 --workload c1 --steps 10 --warmup 2 &&
run c1d64 300 --workload c1d64 --steps 10 --warmup 2 &&
run c3i 300 --workload c3i --steps 20 --warmup 5 --no-cpu-baseline &&
run c2m 300 --workload c2m --no-cpu-baseline &&
run c5m 300 --workload c5m --no-cpu-baseline &&
run c2_k100 300 --workload c2 --k 100 --steps 10 --warmup 3 --no-cpu-baseline &&
run c2_k1000 300 --workload c2 --k 1000 --steps 10 --warmup 3 --no-cpu-baseline &&
run c1_k64 300 --workload c1 --k 64 --steps 10 --warmup 3 --no-cpu-baseline &&
run c1_k100 300 --workload c1 --k 100 --steps 10 --warmup 3 --no-cpu-baseline &&
VFSX_BENCH_BACKEND=gloo run g2_gloo 400 --gpus 2 --workload c2 --steps 5 --warmup 1
;;
b)
run c4 400 --workload c4 --no-cpu-baseline &&
run c4e 400 --workload c4e --no-cpu-baseline &&
run c4i8 400 --workload c4i8 --no-cpu-baseline &&
run c4auto 400 --workload c4auto --no-cpu-baseline &&
run c4eauto 400 --workload c4eauto --no-cpu-baseline &&
run c3b 400 --workload c3b --no-cpu-baseline
;;
esac
