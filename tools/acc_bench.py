"""Time K7 (accumulate) and K6 (percentiles) alone at the cfg-2 size on a synthetic frame."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import _lib as L, pipeline as P
import yond_public_amd.synthetic as S
lib = L.load()
H, W = 3000, 4000
h, w = H // 2, W // 2
o = [torch.empty(4, h, w, device='cuda') for _ in range(4)]
xb = torch.from_numpy(S.synth_noisy(H, W, 4.0, 6.0, 0)[0]).cuda()
lib.yond_box_stats_self1_f32(L.ptr(xb), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.stream())
lib.yond_box_stats_self2_f32(L.ptr(o[2]), h, w, 29, 0, L.ptr(o[3]), L.stream())
lap, mean, var = o[3].reshape(-1), o[0].reshape(-1), o[1].reshape(-1)
ths = P._percentiles(lap, np.linspace(5, 100, 20))
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
q = np.linspace(5, 100, 20)
def nlf_tail():
    occ = P._occupancy(lap, mean, ths, w)
    sel, npk = P._score3_device(occ, ths, q)
    return P._moments(lap, mean, var, sel[1:2])
print("occupancy+score3+moments: %.1f us" % t(nlf_tail), " percentiles: %.1f us" % t(lambda: P._percentiles(lap, q)))
