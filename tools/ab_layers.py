"""Same-box A/B of tools/layer_times.py logs: python tools/ab_layers.py base1 new1 [base2 new2 ...]"""
import re
import sys


def rd(f):
    return [(l[:46].strip(), float(m.group(1))) for l in open(f) for m in [re.search(r"([\d.]+) us$", l)] if m]


logs = [rd(f) for f in sys.argv[1:]]
for i in range(len(logs[0])):
    vals = [lg[i][1] for lg in logs]
    base, new = vals[0::2], vals[1::2]
    print("%-46s %s   delta %+7.1f" % (logs[0][i][0], " ".join("%8.1f" % v for v in vals), sum(new) / len(new) - sum(base) / len(base)))
