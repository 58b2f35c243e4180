"""One pass of the VST / NLE stage kernels at the cfg-2 size, a few times (for rocprofv3 counter passes)."""
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
o8 = torch.empty_like(o[0])
q = np.ascontiguousarray(P.QUANTS)
qp = C.c_void_p(q.ctypes.data)
ws = P._nle_workspace(4 * h * w, x.device)
st = L.stream()
n = 4 * h * w
buf = P._chain_buffers(x.device, 0)
p2d = P.get_p2d((1, 4, h, w), 32)
Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
x4 = torch.empty(Hp, Wp, 4, device='cuda')
mx = torch.empty(1, device='cuda')
out = torch.empty(H, W, device='cuda')
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    lib.yond_box_stats_self_stats_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o8), L.ptr(o[2]), qp, len(q), L.ptr(ws), st)
    lib.yond_nle_threshold_f32(L.ptr(o[2]), n, qp, len(q), 1, L.ptr(ws), st)
    lib.yond_nle_moments_f32(L.ptr(o[2]), L.ptr(o[0]), L.ptr(o[1]), n, L.ptr(ws), st)
    lib.yond_frame_params_f64(L.ptr(ws), None, 0, 959.0, 959.0, 1.03, P.LUT_CAP, L.ptr(buf.prm), L.ptr(buf.t), L.ptr(buf.lut_x), st)
    lib.yond_bias_lut_dev_f64(L.ptr(buf.lut_x), P.LUT_CAP, L.ptr(buf.prm), L.ptr(buf.lut_y), st)
    lib.yond_lut_table_f64(L.ptr(buf.lut_x), L.ptr(buf.lut_y), -1, L.ptr(buf.prm), L.ptr(buf.lut_ws), st)
    lib.yond_pack_vst_norm_dev_f32(L.ptr(x), H, W, L.ptr(x4), p2d[0], p2d[1], p2d[2], p2d[3], 959.0, L.ptr(buf.prm), L.ptr(buf.lut_ws), P.LUT_CAP, L.ptr(mx), st)
    lib.yond_denorm_ivst_unpack_dev_f32(L.ptr(x4), Hp, Wp, p2d[2], p2d[0], h, w, L.ptr(out), 1, 959.0, L.ptr(buf.prm), 1, st)
    lib.yond_box_stats_collab_stats_f32(L.ptr(x), L.ptr(xc), H, W, 29, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[3]), qp, len(q), L.ptr(ws), st)
torch.cuda.synchronize()
