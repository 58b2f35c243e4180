"""Median / avg / min / max duration per kernel from a rocprofv3 --kernel-trace csv dir: python tools/kmedians.py <dir>"""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    d.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
print("%-60s %6s %10s %10s %10s %10s" % ("kernel", "calls", "median_us", "avg_us", "min_us", "max_us"))
for k, v in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-60s %6d %10.1f %10.1f %10.1f %10.1f" % (k[:60], len(v), statistics.median(v), sum(v) / len(v), min(v), max(v)))
