"""Per-kernel means of the counters of one rocprofv3 --pmc pass (counter_collection.csv), with derived MFMA ratios.
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_BUSY_CYCLES / SQ_WAVE_CYCLES quad-cycles... (MI355X_MICROARCH.md): ratios between
kernels are what to read; the raw sums are printed as collected."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", r['Kernel_Name'])[:90]
    acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[name][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
rows = []
for name, c in acc.items():
    n = len(dur[name])
    us = sum(dur[name].values()) / n
    means = {k: sum(v) / len(v) for k, v in c.items()}
    rows.append((us * n, name, n, us, means))
for tot, name, n, us, means in sorted(rows, reverse=True)[:24]:
    extra = ""
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in means and means.get('SQ_BUSY_CYCLES'):
        extra = f"  mfma_busy/sq_busy={means['SQ_VALU_MFMA_BUSY_CYCLES'] / means['SQ_BUSY_CYCLES']:.3f}"
    if 'SQ_LDS_BANK_CONFLICT' in means and means.get('SQ_LDS_IDX_ACTIVE'):
        extra += f"  lds_conflict/active={means['SQ_LDS_BANK_CONFLICT'] / means['SQ_LDS_IDX_ACTIVE']:.3f}"
    print(f"{name:90s} n={n:4d} avg={us:9.1f}us " + " ".join(f"{k}={v:.4g}" for k, v in sorted(means.items())) + extra)
