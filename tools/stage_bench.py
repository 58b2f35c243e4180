"""Stand-alone timings (HIP events, 10 repetitions each) of the VST / NLE kernels of the default path at the cfg-2 size:
the estimator's producers and sweeps, the device parameter chain, K1 (prepared table) and K4."""
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
n = 4 * h * w


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


o8 = torch.empty_like(o[0])
twop = lambda: lib.yond_box_stats_self_stats_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o8), L.ptr(o[2]), qp, len(q), L.ptr(ws), st)
print("self: box kernels + sweep 1 + resolve (yond_box_stats_self_stats_f32): %.1f us" % t(twop))
print("collab: box kernel + sweep 1 + resolve: %.1f us" % t(lambda: lib.yond_box_stats_collab_stats_f32(L.ptr(x), L.ptr(xc), H, W, 29, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[3]), qp, len(q), L.ptr(ws), st)))
print("  of it: box kernels alone (self1 + self2): %.1f us" % t(lambda: (lib.yond_box_stats_self1_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[3]), st),
                                                                         lib.yond_box_stats_self2_f32(L.ptr(o[3]), h, w, 29, 0, L.ptr(o[2]), st))))
print("  of it: sweep 1 alone (yond_nle_stats_f32): %.1f us" % t(lambda: lib.yond_nle_stats_f32(L.ptr(o[2]), L.ptr(o[0]), n, w, qp, len(q), L.ptr(ws), st)))
twop()
def thr():
    lib.yond_nle_stats_f32(L.ptr(o[2]), L.ptr(o[0]), n, w, qp, len(q), L.ptr(ws), st)
    lib.yond_nle_threshold_f32(L.ptr(o[2]), n, qp, len(q), 1, L.ptr(ws), st)
print("sweep 1 + sweep 2 + finish: %.1f us" % t(thr))
print("moments: %.1f us" % t(lambda: lib.yond_nle_moments_f32(L.ptr(o[2]), L.ptr(o[0]), L.ptr(o[1]), n, L.ptr(ws), st)))
# a complete estimate on the workspace (the separate sweep-1 calls above reset it, frame maximum included)
twop()
lib.yond_nle_threshold_f32(L.ptr(o[2]), n, qp, len(q), 1, L.ptr(ws), st)
lib.yond_nle_moments_f32(L.ptr(o[2]), L.ptr(o[0]), L.ptr(o[1]), n, L.ptr(ws), st)
buf = P._chain_buffers(x.device, 0)
def chain():
    lib.yond_frame_params_f64(L.ptr(ws), None, 0, 959.0, 959.0, 1.03, P.LUT_CAP, L.ptr(buf.prm), L.ptr(buf.t), L.ptr(buf.lut_x), st)
    lib.yond_bias_lut_dev_f64(L.ptr(buf.lut_x), P.LUT_CAP, L.ptr(buf.prm), L.ptr(buf.lut_y), st)
    lib.yond_lut_table_f64(L.ptr(buf.lut_x), L.ptr(buf.lut_y), -1, L.ptr(buf.prm), L.ptr(buf.lut_ws), st)
print("device chain: frame params + bias LUT + table: %.1f us" % t(chain))
print("  of it: frame params %.1f us, bias LUT %.1f us, table %.1f us" % (
    t(lambda: lib.yond_frame_params_f64(L.ptr(ws), None, 0, 959.0, 959.0, 1.03, P.LUT_CAP, L.ptr(buf.prm), L.ptr(buf.t), L.ptr(buf.lut_x), st)),
    t(lambda: lib.yond_bias_lut_dev_f64(L.ptr(buf.lut_x), P.LUT_CAP, L.ptr(buf.prm), L.ptr(buf.lut_y), st)),
    t(lambda: lib.yond_lut_table_f64(L.ptr(buf.lut_x), L.ptr(buf.lut_y), -1, L.ptr(buf.prm), L.ptr(buf.lut_ws), st))))
print("device chain as ONE launch (yond_frame_chain_f64): %.1f us" % t(lambda: lib.yond_frame_chain_f64(
    L.ptr(ws), None, 0, 959.0, 959.0, 1.03, P.LUT_CAP, L.ptr(buf.prm), L.ptr(buf.t), L.ptr(buf.lut_x), L.ptr(buf.lut_y), L.ptr(buf.lut_ws), st)))
torch.cuda.synchronize()
prm = buf.prm.cpu().numpy()
print("  parameter block: flags %d, K %.4f, sigma %.4f, %d knots" % (int(prm[P.PRM['flags']]), prm[P.PRM['gain']], prm[P.PRM['sigma']], int(prm[P.PRM['lut_n']])))
assert int(prm[P.PRM['flags']]) == 0
p2d = P.get_p2d((1, 4, h, w), 32)
Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
x4 = torch.empty(Hp, Wp, 4, device='cuda')
mx = torch.empty(1, device='cuda')
out = torch.empty(H, W, device='cuda')
print("K1 (prepared table, constants from the block): %.1f us" % t(lambda: lib.yond_pack_vst_norm_dev_f32(L.ptr(x), H, W, L.ptr(x4), p2d[0], p2d[1], p2d[2], p2d[3], 959.0, L.ptr(buf.prm), L.ptr(buf.lut_ws), P.LUT_CAP, L.ptr(mx), st)))
print("K1 (the chain's kernel: folded coefficients, runs in registers): %.1f us" % t(lambda: lib.yond_pack_vst_norm_chain_f32(L.ptr(x), H, W, L.ptr(x4), p2d[0], p2d[1], p2d[2], p2d[3], 959.0, L.ptr(buf.prm), L.ptr(buf.lut_ws), P.LUT_CAP, L.ptr(mx), st)))
lut = P.get_bias(np.float32(noisy.max()) * np.float32(959.0), np.float64(6.0), np.float64(4.0), device=x.device)
lo, hi = P.vst_scalar(0, np.float64(6.0), np.float64(4.0)), P.vst_scalar(959.0, np.float64(6.0), np.float64(4.0))
print("K1 (host-side entry: table derived per workgroup): %.1f us" % t(lambda: lib.yond_pack_vst_norm_f32(L.ptr(x), H, W, L.ptr(x4), p2d[0], p2d[1], p2d[2], p2d[3], 1, 959.0, 4.0, 6.0, float(lo), float(hi), L.ptr(lut.x), L.ptr(lut.y), len(lut), L.ptr(mx), st)))
print("K4: %.1f us" % t(lambda: lib.yond_denorm_ivst_unpack_dev_f32(L.ptr(x4), Hp, Wp, p2d[2], p2d[0], h, w, L.ptr(out), 1, 959.0, L.ptr(buf.prm), 1, st)))
print("SimpleNLF self end to end (host-side chain, incl. its host sync): %.1f us" % t(lambda: P.SimpleNLF(x, k=29, setting={'mode': 'self'}), 20))
