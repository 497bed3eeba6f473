#!/bin/bash
# round-3 rocprofv3 evidence (GPU box): kernel trace + FETCH/WRITE passes per workload, SQ passes for c2/c3/c4
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() { # tag, then bench args
  local TAG=$1; shift
  local OUT=gpurun_out/prof_$TAG
  rm -rf $OUT && mkdir -p $OUT
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || { echo "trace failed $TAG"; tail -3 $OUT/trace.log; return 1; }
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline "$@" --steps 2 --warmup 1 --queries-per-step $QPS > $OUT/pmc_fetch.log 2>&1 || { echo "fetch failed $TAG"; return 1; }
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline "$@" --steps 2 --warmup 1 --queries-per-step $QPS > $OUT/pmc_write.log 2>&1 || { echo "write failed $TAG"; return 1; }
  python3 tools/summarize_profile.py $TAG r03 | tail -4
}
QPS=16 prof c2 --workload c2 --steps 10 --warmup 2 &&
QPS=16 prof c3 --workload c3 --steps 10 --warmup 2 &&
QPS=8 prof c5 --workload c5 --steps 4 --warmup 1 &&
QPS=4096 prof c4 --workload c4 --steps 5 --warmup 1 &&
QPS=4096 prof c4e --workload c4e --steps 5 --warmup 1 &&
QPS=4096 prof c4i8 --workload c4i8 --steps 5 --warmup 1 &&
QPS=4096 prof c4auto --workload c4auto --steps 5 --warmup 1
