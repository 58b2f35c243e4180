"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM bytes per kernel.
gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports half of the bytes of
wide coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-byte stores.
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


def collect(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc


def short(name):
    m = re.search(r"conv_mfma_kernel<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(',')]
        return f"conv_mfma_kernel<{a[0]},{a[1]},{a[2]},{a[3]},{a[4]}>"
    m = re.search(r"conv_split_kernel<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(',')]
        k1 = len(a) > 8 and a[8] == 'true'                         # the decoder's 1x1 GEMM form
        return f"conv_split_kernel<{'k1' if k1 else a[0]},{a[2]},{a[4]}>"   # stride, channel-tile width, parts (bench.py's tag)
    m = re.search(r"conv_wino_kernel<\s*(\d+)", name)
    if m:
        return f"conv_wino_kernel<{m.group(1)}>"
    return name.split('(')[0]


fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in fetch:
    fs = sum(fetch[k]) / len(fetch[k]) * 1024 * 2          # KiB -> B, x2 (gfx950 wide-read under-count)
    ws = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1) * 1024
    e = out.setdefault(short(k), {"launches": 0, "fetch": 0.0, "write": 0.0})
    n = len(fetch[k])
    e["fetch"] = (e["fetch"] * e["launches"] + fs * n) / (e["launches"] + n)
    e["write"] = (e["write"] * e["launches"] + ws * n) / (e["launches"] + n)
    e["launches"] += n
for k, e in out.items():
    e["hbm_bytes_per_launch"] = round(e["fetch"] + e["write"])
    e["fetch_bytes_per_launch_x2_corrected"] = round(e.pop("fetch"))
    e["write_bytes_per_launch"] = round(e.pop("write"))
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, e in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
    print(f"{k[:60]:60s} n={e['launches']:3d} fetch={e['fetch_bytes_per_launch_x2_corrected']/1e6:9.1f} MB write={e['write_bytes_per_launch']/1e6:9.1f} MB")
