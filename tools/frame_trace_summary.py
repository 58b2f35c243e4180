import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [re.sub(r"\(.*", "", r['Kernel_Name'])[:70] for r in rows]
k1 = [i for i, n in enumerate(names) if n.startswith('pack_vst_')]         # (K1: pack_vst_chain_kernel on the default path)
a, b = k1[-2], k1[-1]
cnt = collections.Counter(names[a:b])
print("launches between the last two K1 launches: %d" % (b - a))
for n, c in cnt.most_common():
    print("%3d  %s" % (c, n))
