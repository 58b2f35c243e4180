"""Inline-asm memory instructions are invisible to the compiler's hazard recognizer.  This scan compiles every translation unit of
csrc/ to gfx950 assembly and looks, in front of each inline-asm style VMEM instruction that reads scalar registers
(global_load_lds_* with an SGPR base, buffer_load_* with a descriptor), for a vector-ALU write to one of those scalars
(v_readfirstlane / v_readlane / v_cmp into an SGPR pair) fewer than five wait states earlier -- the gfx9 rule
"VALU writes SGPR -> VMEM reads that SGPR: 5 wait states".  (How the rule was met: csrc/train.hip, the weight-gradient kernel.)
    python tools/asm_hazard_scan.py [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "yond_public_amd", "csrc", "*.hip")))
total = bad = 0
for src in srcs:
    out = f"/tmp/hazard_{os.path.basename(src)}.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include",
                        f"-I{ROOT}/yond_public_amd/csrc", "-S", "--cuda-device-only", src, "-o", out], capture_output=True, text=True)
    if r.returncode:
        print(src, "did not compile:", r.stderr[-500:])
        sys.exit(2)
    ins = [l.strip() for l in open(out)]
    ins = [l for l in ins if l and not l.startswith((".", ";", "//")) and not l.endswith(":")]
    n = 0
    for i, t in enumerate(ins):
        if not t.startswith(("global_load_lds", "buffer_load", "buffer_store")):
            continue
        regs = set()
        for m in re.finditer(r"s\[(\d+):(\d+)\]", t):
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
        if not regs:
            continue
        n += 1
        ws, j = 0, i - 1
        while j >= 0 and ws < 5:
            p = ins[j]
            if p.startswith("s_nop"):
                ws += int(p.split()[1]) + 1
            else:
                d = None
                if p.startswith(("v_readfirstlane", "v_readlane")):
                    d = re.match(r"\S+ s(\d+)", p)
                    hit = d and int(d.group(1)) in regs
                elif p.startswith("v_cmp") or p.startswith("v_add_co") or p.startswith("v_sub_co"):
                    d = re.search(r"s\[(\d+):(\d+)\]", p.split(",")[0] if p.startswith("v_cmp") else p.split(",")[1])
                    hit = d and any(k in regs for k in range(int(d.group(1)), int(d.group(2)) + 1))
                else:
                    hit = False
                if hit:
                    bad += 1
                    print(f"{os.path.basename(src)}: {p}   ->   {t}   ({ws} wait states)")
                ws += 1
            j -= 1
    total += n
    print(f"{os.path.basename(src):28s} {n:5d} scalar-operand VMEM instructions checked", flush=True)
print(f"{total} instructions, {bad} with a vector-ALU write to their scalars inside five wait states")
sys.exit(1 if bad else 0)
