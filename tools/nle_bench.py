"""Time the NLE kernels at the cfg-2 size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import _lib as L
lib = L.load()
H, W = 3000, 4000
h, w = H // 2, W // 2
x = torch.rand(H, W, device='cuda')
o = [torch.empty(4, h, w, device='cuda') for _ in range(4)]

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for k2 in (19, 29, 9):
    print("self1 k2=%d: %.1f us" % (k2, t(lambda: lib.yond_box_stats_self1_f32(L.ptr(x), H, W, 29, k2, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.stream()))))
print("self2: %.1f us" % t(lambda: lib.yond_box_stats_self2_f32(L.ptr(o[2]), h, w, 29, 0, L.ptr(o[3]), L.stream())))
print("collab: %.1f us" % t(lambda: lib.yond_box_stats_collab_f32(L.ptr(x), L.ptr(x), H, W, 29, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.stream())))
