#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/r01/<tag>_summary.{md,json}.
HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads, so the
read side is doubled.  Separate --pmc passes, one counter each."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = f"gpurun_out/prof_{tag}"
dst = f"profiles/{rnd}"
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = newest(f"{src}/trace/*/*_kernel_stats.csv")
shutil.copy(stats, f"{dst}/{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))


def pmc(kind):
    f = newest(f"{src}/pmc_{kind}/*/*_counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = pmc("fetch"), pmc("write")
bench_line = [l for l in open(f"{src}/trace.log") if l.startswith("{")][-1]
bench = json.loads(bench_line)
out = {"bench_line_under_profiler": bench, "kernels": []}
md = [f"# rocprofv3 summary `{tag}` ({rnd})", "",
      "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline "
      f"--workload {tag}` (steps/warmup as in tools/{rnd}_profiles.sh); PMC passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` (one run each).",
      "", "| kernel | calls | avg µs | % time | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM bytes/launch (2·fetch+write) |",
      "|---|---|---|---|---|---|---|"]
for r in rows:
    name = r["Name"]
    import re
    m = re.search(r"(\w+)(<.*>)?\(", name.replace("(anonymous namespace)::", ""))
    short = name if not m else (m.group(1) + (m.group(2) or ""))
    short = short.replace("vfsx::", "")
    f = fetch.get(name)
    w = write.get(name)
    traffic = None if f is None or w is None else int(2 * f * 1024 + w * 1024)
    out["kernels"].append({"name": name, "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                           "pct": float(r["Percentage"]), "fetch_kib": f, "write_kib": w, "hbm_bytes": traffic})
    md.append(f"| `{short[:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} | "
              f"{'' if f is None else round(f,1)} | {'' if w is None else round(w,1)} | {'' if traffic is None else traffic} |")
dom = bench["roofline"]["kernel"]
scan = [k for k in out["kernels"] if dom in k["name"]][0]
if bench["roofline"]["bound"] == "mfma":
    flops = bench["roofline"]["algorithmic_flops_per_launch"]
    peak, unit = bench["roofline"]["peak"], bench["roofline"]["unit"]
    md += ["", f"Dominant kernel: {dom}. Algorithmic flops/launch {flops:.4g}; avg duration {scan['avg_ns']/1e6:.2f} ms -> "
           f"{flops/scan['avg_ns']/1e3:.1f} {unit} = {flops/scan['avg_ns']/1e3/peak*100:.1f}% of the {peak/1000:g} P{unit[1:]} dense MFMA peak "
           f"for this operand type; "
           f"HBM bytes/launch {scan['hbm_bytes']} (mirror + query re-reads that miss L2/MALL).",
           "", "bench.py line from the same (profiled) run:", "", "```", bench_line.strip(), "```"]
else:
    algo = bench["roofline"]["algorithmic_bytes_per_launch"]
    md += ["", f"Dominant kernel: scan. Algorithmic bytes/launch {algo}; measured HBM bytes/launch {scan['hbm_bytes']} "
           f"(ratio {scan['hbm_bytes']/algo:.4f}); avg duration {scan['avg_ns']/1e3:.1f} µs -> "
           f"{algo/scan['avg_ns']:.1f} GB/s algorithmic = {algo/scan['avg_ns']/8000*100:.1f}% of the 8 TB/s HBM3E peak.",
           "", "bench.py line from the same (profiled) run:", "", "```", bench_line.strip(), "```"]
out["scan_traffic_bytes_per_launch"] = scan["hbm_bytes"]
open(f"{dst}/{tag}_summary.md", "w").write("\n".join(md) + "\n")
json.dump(out, open(f"{dst}/{tag}_summary.json", "w"), indent=1)
# bench.py reads this file for roofline.traffic
tf = f"{dst}/traffic.json"
t = json.load(open(tf)) if os.path.exists(tf) else {}
t[tag] = {"scan_hbm_bytes_per_launch": scan["hbm_bytes"], "fetch_kib": scan["fetch_kib"], "write_kib": scan["write_kib"],
          "correction": "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE reports half of a wide streaming read)"}
json.dump(t, open(tf, "w"), indent=1)
print("\n".join(md))
