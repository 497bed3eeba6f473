"""Per-kernel durations and launch gaps from a rocprofv3 --kernel-trace run (its rocpd .db output): name, calls,
average / min / max ns, and the median gap between consecutive dispatches.  usage: kernel_times.py results.db"""
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
for name, n, avg, lo, hi in db.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"):
    print(f"{name[:90]:90s} {n:7d} avg {avg:10.0f} min {lo:8d} max {hi:8d} ns")
ks = db.execute("select start, end from kernels order by start").fetchall()
gaps = [b[0] - a[1] for a, b in zip(ks, ks[1:])]
if gaps:
    print("median gap between dispatches", statistics.median(gaps), "ns")
