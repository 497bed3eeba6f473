#!/bin/bash
# C4 (bf16 mirror): query tiles per XCD in the prefilter's mapping -- kernel time and FETCH_SIZE per launch
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-c4}; shift
for q in "$@"; do
  export VFSX_MFMA_QPX=$q
  O=gpurun_out/qpx_${W}_$q; rm -rf $O; mkdir -p $O
  timeout -k 10 300 python3 bench.py --workload $W --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc -- python3 bench.py --no-cpu-baseline --workload $W --steps 2 --warmup 1 > $O/pmc.log 2>&1 || { echo "pmc failed"; exit 1; }
  python3 - "$O" "$q" <<'PY'
import sys,json,glob,csv,collections
o,q=sys.argv[1],sys.argv[2]
d=json.loads([l for l in open(o+'/bench.json') if l.startswith('{')][-1])
f=glob.glob(o+'/pmc/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append(float(r['Counter_Value']))
k=[k for k in agg if 'mfma_prefilter256' in k][0]
v=sum(agg[k])/len(agg[k])
print(f"qpx={q} kernel_ms={d['roofline']['avg_kernel_ms']} frac={d['roofline']['frac']} value={d['value']} FETCH_SIZE={v*1024/1e9:.1f} GB raw, x2 = {2*v*1024/1e9:.1f} GB per launch")
PY
done
