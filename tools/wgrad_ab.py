"""Weight-gradient kernels side by side at the reference's training shapes (batch 64 of [4][128][128] patches, nf 32): the
fp32-input MFMA kernel (yond_conv_wgrad_ws_f32) against the split-operand fp16-MFMA kernel (yond_conv_wgrad_split_f32), per level."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import _lib as L
lib = L.load()
N = 64
for lvl in range(5):
    C, H = 32 << lvl, 128 >> lvl
    x = torch.randn(N, H, H, C, device='cuda')
    dy = torch.randn(N, H, H, C, device='cuda') * 0.3
    dw = torch.empty(9, C, C, device='cuda'); dw2 = torch.empty_like(dw)
    n0 = int(lib.yond_conv_wgrad_ws_bytes(N, H, H, C, H, H, C, 0, 1)); n1 = int(lib.yond_conv_wgrad_split_ws_bytes(N, H, H, C, C))
    ws0 = torch.empty(n0 // 4, device='cuda'); ws1 = torch.empty(max(n1, 4) // 4, device='cuda')
    st = torch.zeros(1, dtype=torch.int32, device='cuda')
    f0 = lambda: lib.yond_conv_wgrad_ws_f32(L.ptr(x), L.ptr(dy), N, H, H, C, H, H, C, 0, 1, L.ptr(dw), L.ptr(ws0), n0, L.stream())
    f1 = lambda: lib.yond_conv_wgrad_split_f32(L.ptr(x), L.ptr(dy), N, H, H, C, C, L.ptr(dw2), 0, L.ptr(ws1), n1, L.ptr(st), L.stream())

    def t(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    a, b = t(f0), t(f1)
    gf = 2 * 9 * C * C * N * H * H / 1e9
    err = float((dw2 - dw).abs().max() / dw.abs().max())
    print(f"level {lvl}: C {C:3d} {H:3d}x{H:<3d} {gf:5.1f} GFLOP  fp32-MFMA {a:7.1f} us ({gf / a * 1e3:6.1f} TF/s)   split fp16-MFMA {b:7.1f} us ({gf / b * 1e3:6.1f} TF/s)"
          f"   x{a / b:.2f}   ws {n1 / 2**20:.0f} MiB  max rel diff {err:.1e}  status {int(st.item())}", flush=True)
