"""Time the round-2 estimator kernels (fused box statistics, two-sweep selection, moments) and K1 / K4 at the cfg-2 size."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import _lib as L
from yond_public_amd import pipeline as P
import yond_public_amd.synthetic as S

lib = L.load()
H, W = 3000, 4000
h, w = H // 2, W // 2
noisy, clean = S.synth_noisy(H, W, 4.0, 6.0, 0)
x = torch.from_numpy(noisy).cuda()
xc = torch.from_numpy(clean).cuda()
o = [torch.empty(4, h, w, device='cuda') for _ in range(4)]
q = np.ascontiguousarray(P.QUANTS)
qp = C.c_void_p(q.ctypes.data)
ws = P._nle_workspace(4 * h * w, x.device)
st = L.stream()


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fused = lambda: lib.yond_box_stats_self_fused_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), qp, len(q), L.ptr(ws), st)
print("fused self (memset + kernel + resolve): %.1f us" % t(fused))
print("fused collab: %.1f us" % t(lambda: lib.yond_box_stats_collab_fused_f32(L.ptr(x), L.ptr(xc), H, W, 29, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[3]), qp, len(q), L.ptr(ws), st)))
o8 = torch.empty_like(o[0])
twop = lambda: lib.yond_box_stats_self_stats_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o8), L.ptr(o[2]), qp, len(q), L.ptr(ws), st)
print("two-pass self (memset + self1 + self2 with statistics + resolve): %.1f us" % t(twop))
print("two-pass collab: %.1f us" % t(lambda: lib.yond_box_stats_collab_stats_f32(L.ptr(x), L.ptr(xc), H, W, 29, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[3]), qp, len(q), L.ptr(ws), st)))
fused()
n = 4 * h * w
print("stats sweep (stand-alone): %.1f us" % t(lambda: lib.yond_nle_stats_f32(L.ptr(o[2]), L.ptr(o[0]), n, w, qp, len(q), L.ptr(ws), st)))
def thr():
    fused_or_stats()
    lib.yond_nle_threshold_f32(L.ptr(o[2]), n, qp, len(q), 1, L.ptr(ws), st)
fused_or_stats = lambda: lib.yond_nle_stats_f32(L.ptr(o[2]), L.ptr(o[0]), n, w, qp, len(q), L.ptr(ws), st)
print("stats + collect + final: %.1f us" % t(thr))
off = P._nle_layout()
base = ws.data_ptr()
print("moments: %.1f us" % t(lambda: lib.yond_nlf_moments_f32(L.ptr(o[2]), L.ptr(o[0]), L.ptr(o[1]), n, C.c_void_p(base + off[1] + 8), C.c_void_p(base + off[2]), st)))
print("old self1+self2: %.1f us" % t(lambda: (lib.yond_box_stats_self1_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[3]), st),
                                              lib.yond_box_stats_self2_f32(L.ptr(o[3]), h, w, 29, 0, L.ptr(o[2]), st))))
for box in ('two-pass', 'one-pass', 'plain'):
    print("SimpleNLF self  box=%-8s end to end (incl. host sync): %.1f us" % (box, t(lambda: P.SimpleNLF(x, k=29, setting={'mode': 'self'}, box=box), 20)))
for box in ('two-pass', 'one-pass', 'plain'):
    print("SimpleNLF collab box=%-8s end to end (incl. host sync): %.1f us" % (box, t(lambda: P.SimpleNLF(x, xc, k=29, setting={'mode': 'collab'}, box=box), 20)))

# K1 / K4
from yond_public_amd import pipeline as PP
lut = PP.get_bias(np.float32(noisy.max()) * np.float32(959.0), np.float64(6.0), np.float64(4.0), device=x.device)
lo, hi = PP.vst_scalar(0, np.float64(6.0), np.float64(4.0)), PP.vst_scalar(959.0, np.float64(6.0), np.float64(4.0))
p2d = PP.get_p2d((1, 4, h, w), 32)
Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
x4 = torch.empty(Hp, Wp, 4, device='cuda')
mx = torch.empty(1, device='cuda')
out = torch.empty(H, W, device='cuda')
print("K1: %.1f us" % t(lambda: lib.yond_pack_vst_norm_f32(L.ptr(x), H, W, L.ptr(x4), p2d[0], p2d[1], p2d[2], p2d[3], 1, 959.0, 4.0, 6.0, float(lo), float(hi), L.ptr(lut.x), L.ptr(lut.y), len(lut), L.ptr(mx), st)))
print("K4: %.1f us" % t(lambda: lib.yond_denorm_ivst_unpack_f32(L.ptr(x4), Hp, Wp, p2d[2], p2d[0], h, w, L.ptr(out), 1, 959.0, 4.0, 6.0, float(lo), float(hi), 1, st)))
