"""The weight-gradient kernels of csrc/train.hip feed their MFMAs from inline-asm buffer loads with COUNTED waits: correct only
while (1) nothing of the kernel lives in scratch (a spill reload is a VMEM operation the counts do not know), (2) no instruction
touches a load's destination register before the s_waitcnt that covers it.  This scan compiles the file to gfx950 assembly and
checks, for every wgrad_rows_kernel instantiation: .private_segment_fixed_size 0, .vgpr_spill_count 0, no scratch_ instruction,
and -- replaying the in-order vmcnt bookkeeping over the kernel's instruction stream -- no MFMA that reads a register whose
buffer_load_dword is still outstanding.  The split-operand weight-gradient kernels (csrc/wgrad_split.hip) are checked for
spills only (their loads are compiler-visible).
    python tools/asm_pipeline_scan.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm_of(name):
    src = os.path.join(ROOT, "yond_public_amd", "csrc", name)
    out = f"/tmp/pipe_{name}.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include",
                        f"-I{ROOT}/yond_public_amd/csrc", "-S", "--cuda-device-only", src, "-o", out], capture_output=True, text=True)
    if r.returncode:
        print(src, "did not compile:", r.stderr[-500:])
        sys.exit(2)
    return open(out).read()


def kernels(text, prefix):
    """{mangled name: body lines} of the kernels whose demangled name starts with prefix, and their metadata blocks."""
    out = {}
    for m in re.finditer(r"^(_Z\w*%s\w*):\s*;?.*$" % prefix, text, re.M):
        name = m.group(1)
        end = text.find(".end_amdhsa_kernel", m.end())
        body = text[m.end():text.rfind("s_endpgm", m.end(), end) if text.rfind("s_endpgm", m.end(), end) > 0 else end]
        out[name] = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.strip().endswith(":")]
    return out


def meta(text, name, key):
    i = text.find(f".name:           {name}")
    blk = text[max(0, text.rfind("  - .agpr_count", 0, i)):text.find("  - .agpr_count", i) if text.find("  - .agpr_count", i) > 0 else len(text)]
    m = re.search(r"\.%s:\s*(\d+)" % key, blk)
    return int(m.group(1)) if m else None


def regs_of(tok):
    r = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        r.add(int(m.group(1)))
    return r


bad = 0
t = asm_of("train.hip")
ks = kernels(t, "wgrad_rows_kernel")
assert len(ks) >= 5, list(ks)
nload = 0
for name, ins in ks.items():
    for key in ("private_segment_fixed_size", "vgpr_spill_count"):
        v = meta(t, name, key)
        if v != 0:
            bad += 1
            print(f"{name}: .{key} = {v}")
    pending = []                                   # destination registers of outstanding VMEM operations, in issue order
    for l in ins:
        op = l.split()[0]
        if op.startswith("scratch_"):
            bad += 1
            print(f"{name}: {l}")
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n < len(pending) else pending
                if n == 0:
                    pending = []
            continue
        if op.startswith(("buffer_load", "global_load")):
            pending.append(regs_of(l.split(",")[0]))
            nload += 1
            continue
        if op.startswith(("buffer_store", "global_store", "global_atomic", "buffer_atomic")):
            pending.append(set())                  # counts in vmcnt, has no destination
            continue
        # the consumers that matter are the MFMAs (the manual, counted waits stand in front of them; the compiler knows the asm
        # statements' outputs and keeps other values out of those registers by itself)
        if op.startswith("v_mfma"):
            hot = set().union(*pending) if pending else set()
            srcs = l.split(None, 1)[1].split(",")[1:3]
            if hot and regs_of(",".join(srcs)) & hot:
                bad += 1
                print(f"{name}: an MFMA reads a register whose load is still outstanding: {l}")
t2 = asm_of("wgrad_split.hip")
ks2 = kernels(t2, "wgrad_split_kernel")
assert len(ks2) >= 5, list(ks2)
for name in ks2:
    for key in ("private_segment_fixed_size", "vgpr_spill_count"):
        v = meta(t2, name, key)
        if v != 0:
            bad += 1
            print(f"{name}: .{key} = {v}")
print(f"{len(ks)} wgrad_rows_kernel + {len(ks2)} wgrad_split_kernel instantiations, {nload} loads replayed, {bad} findings")
sys.exit(1 if bad else 0)
