"""Per-launch summary of the training step's kernels from a rocprofv3 kernel trace: python tools/train_trace.py trace.csv [name-substring]
groups the launches of one step by (kernel, grid) and prints mean duration and count."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else "wgrad"
g = defaultdict(list)
for r in rows:
    if sub in r["Kernel_Name"]:
        key = (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"], r["Grid_Size_Y"])
        g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:60s} grid {k[1]:>8s} x {k[2]:>5s}  n {len(v):4d}  mean {sum(v) / len(v):9.1f} us  total {sum(v) / 1e3:8.2f} ms")
    tot += sum(v)
print(f"total {tot / 1e3:.2f} ms over all recorded steps")
