"""Print the top kernels of a rocprofv3 --stats run: python tools/kstats.py <dir> [steps]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%-64s calls=%4s avg_us=%8.1f ms/step=%7.3f pct=%5.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                 float(r["TotalDurationNs"]) / 1e6 / steps, float(r["Percentage"])))
print("total kernel ms per step", tot / 1e6 / steps)
