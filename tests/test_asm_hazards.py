"""CPU: the inline-asm memory instructions of the weight-gradient kernel are invisible to the compiler's hazard recognizer; this
compiles csrc/train.hip to gfx950 assembly (hipcc cross-compiles without a GPU) and checks that no buffer load reads a scalar
register the vector ALU wrote fewer than five wait states earlier (tools/asm_hazard_scan.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_kernels_have_no_valu_sgpr_vmem_hazard():
    src = os.path.join(ROOT, "yond_public_amd", "csrc", "train.hip")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_hazard_scan.py"), src], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    n = int(last.split()[0])
    assert n >= 100 and " 0 with a vector-ALU write" in last, last          # (the scan really saw the asm loads)
