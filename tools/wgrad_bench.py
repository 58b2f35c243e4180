"""Weight-gradient kernel alone, same process: python tools/wgrad_bench.py  -- the five 3x3 shapes of GuidedResUnet nf 32 at batch 64 x 128 x 128,
with the workspace (partial sums stored, a second kernel adds them) and without (atomics)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yond_public_amd import _lib as L

lib = L.load()
dev = torch.device('cuda', 0)
N = 64
for lvl in range(5):
    C, H = 32 << lvl, 128 >> lvl
    x = torch.randn(N, H, H, C, device=dev)
    dy = torch.randn(N, H, H, C, device=dev)
    dw = torch.empty(9, C, C, device=dev)
    db = torch.empty(C, device=dev)
    nws = int(lib.yond_conv_wgrad_ws_bytes(N, H, H, C, H, H, C, 0, 1))
    ws = torch.empty(nws // 4, device=dev)
    out = []
    for use_db in (False,):
        for use_ws in (False, True):
            def run():
                L.check(lib.yond_conv_wgrad_ws_f32(L.ptr(x), L.ptr(dy), N, H, H, C, H, H, C, 0, 1, L.ptr(dw),
                                                   L.ptr(ws) if use_ws else None, nws if use_ws else 0, L.stream()), "wgrad")
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            out.append(f"ws={int(use_ws)}: {e0.elapsed_time(e1) * 100:7.1f} us")
    gf = 2 * 9 * C * C * N * H * H / 1e9
    print(f"C {C:4d} {H:3d}x{H:<3d} ({gf:.1f} GFLOP, ws {nws / 1e6:.1f} MB)  " + "  ".join(out), flush=True)
