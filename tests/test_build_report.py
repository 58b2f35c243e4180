"""CPU: the compiler's per-kernel resource remarks, recorded by yond_public_amd/build.py beside every object: no kernel the default
inference flow launches (split-plane / planes-of-4 data flow: LDS-DMA or split-plane input, split-plane output, decoder GEMMs, the
stride-2 layers, the resident-weight level-0 kernels) and no weight-gradient kernel may spill a vector register or use scratch."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(name):
    return [int(v) for _, v in re.findall(r"L([ib])(\d+)E", name)]


def test_flow_kernels_do_not_spill():
    from yond_public_amd import build as B
    B.build_lib(verbose=False)                       # (reused when the sources are unchanged; writes the reports when it compiles)
    rep = B.resource_report()
    if not rep:
        import pytest
        pytest.skip("a shipped library without its object directory or link-time report: nothing to check here")
    assert len(rep) >= 100, len(rep)
    split = [r for r in rep if "conv_split_kernel" in r["name"]]
    assert len(split) >= 30
    flow, other = [], []
    for r in split:
        a = _args(r["name"])                         # STRIDE, TH, TN, MW, PARTS, NWB, PRE, O4, K1, ISPM, OSP, S2, D2
        a += [0] * (13 - len(a))
        (flow if (a[9] != 0 or a[10] or a[12]) else other).append(r)
    assert len(flow) >= 14
    bad = [r for r in flow if r.get("vgpr_spill", 0) or r.get("scratch", 0)]
    assert not bad, bad
    # the fp16 path (BASELINE cfg 5) runs a guided net's whole forward on the flow's H-ONLY instantiations (PARTS 1: tests/test_hip_net.py
    # asserts that its forward launches nothing else): every one of them is in `flow` above -- none spills
    half_flow = [r for r in flow if (_args(r["name"]) + [0] * 13)[4] == 1]
    assert len(half_flow) >= 15, len(half_flow)
    wg = [r for r in rep if "wgrad_split_kernel" in r["name"] or "wgrad_rows_kernel" in r["name"]]
    assert len(wg) >= 10 and not [r for r in wg if r.get("vgpr_spill", 0) or r.get("scratch", 0)], wg
    # the [N][H][W][C]-path instantiations (training forward / data gradients, UNetSeeInDark's unfused layers) keep a few spilled
    # registers (prologue stores + reloads at the tile's end): bounded here so that a regression shows
    assert max(r.get("vgpr_spill", 0) for r in other) <= 24, [r for r in other if r.get("vgpr_spill", 0) > 24]
