"""The box-statistics kernels alone (for rocprofv3 counter passes): self stage 1 + 2 and collab at the cfg-2 size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import _lib as L
import yond_public_amd.synthetic as S

lib = L.load()
H, W = 3000, 4000
h, w = H // 2, W // 2
noisy, clean = S.synth_noisy(H, W, 4.0, 6.0, 0)
x = torch.from_numpy(noisy).cuda()
xc = torch.from_numpy(clean).cuda()
o = [torch.empty(4, h, w, device='cuda') for _ in range(4)]
st = L.stream()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    lib.yond_box_stats_self1_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[3]), st)
    lib.yond_box_stats_self2_f32(L.ptr(o[3]), h, w, 29, 0, L.ptr(o[2]), st)
    lib.yond_box_stats_collab_f32(L.ptr(x), L.ptr(xc), H, W, 29, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), st)
torch.cuda.synchronize()
