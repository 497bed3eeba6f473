#!/bin/bash
# one rocprofv3 --pmc pass over bench.py (run on the GPU box through gpurun)
#   $1 = tag, $2 = kernel-name substring, $3 = "COUNTER COUNTER ..." (<= 8 SQ / 4 TCC / 2 GRBM), rest = bench args
set -e
TAG=$1; KERN=$2; CTRS=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline "$@" > $OUT.log 2>&1
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int); dur = collections.defaultdict(float)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            dur[r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(acc):
    print(f"{k:32s} per-launch {acc[k]/n[k]:.6g}  launches {n[k]}  avg_ns {dur[k]/n[k]:.0f}")
PY
