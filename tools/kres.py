"""Per-kernel register / spill / LDS summary of a HIP source: python tools/kres.py csrc/file.hip [extra hipcc flags]
(hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel)."""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = open(sys.argv[2]).read() if len(sys.argv) > 2 and sys.argv[2].endswith(".txt") else subprocess.run(cmd, capture_output=True, text=True).stderr
open("/tmp/kres_last.txt", "w").write(out)
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    print("%-100s vgpr %3d agpr %3d spill %3d scratch %4d sgpr %3d sspill %3d occ %d" % (
        r["name"][:100], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1), r.get("ScratchSize", -1),
        r.get("SGPRs", -1), r.get("SGPRs Spill", -1), r.get("Occupancy", -1)))
if not rows:
    print(out[-3000:])
