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


def test_wgrad_kernels_keep_their_pipeline_assumptions():
    """the counted-wait pipeline of wgrad_rows_kernel holds only without scratch traffic and with every MFMA behind
    the wait that covers its operands; the split-operand weight-gradient kernels must not spill (tools/asm_pipeline_scan.py)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_pipeline_scan.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert "5 wgrad_rows_kernel + 5 wgrad_split_kernel" in last and last.endswith(" 0 findings"), last
    assert int(last.split(",")[1].split()[0]) >= 100, last                   # (the replay really saw the asm loads)
