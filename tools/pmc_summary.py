"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel dispatch (one line each)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (r['Dispatch_Id'], r['Kernel_Name'][:48])
    e = agg.setdefault(k, {'t': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
    e[r['Counter_Name']] = float(r['Counter_Value'])
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for (did, kn), c in agg.items():
    if pat not in kn:
        continue
    t = c.pop('t')
    print(f"{did:>4s} {kn:48s} {t:9.1f}us " + " ".join(f"{k}={v:.4g}" for k, v in c.items()))
