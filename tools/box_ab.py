"""A/B of the box-statistics kernels: this build against an older library (default tools/probe/libyond_hip_r3a.so):
bit differences of the maps on several geometries, then timings at the cfg-2 size."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import _lib as L
import yond_public_amd.synthetic as S

new = L.load()
old = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "probe", "libyond_hip_r3a.so"))
for name in ("yond_box_stats_self1_f32", "yond_box_stats_self2_f32", "yond_box_stats_collab_f32"):
    f = getattr(new, name)
    g = getattr(old, name)
    g.argtypes, g.restype = f.argtypes, f.restype
st = L.stream()


def ulps(a, b):
    ia, ib = a.view(torch.int32).to(torch.int64), b.view(torch.int32).to(torch.int64)
    d = (ia - ib).abs()
    return int(d.max()), float((d != 0).double().mean())


def run(lib, x, xc, H, W, k, k2, tw):
    h, w = H // 2, W // 2
    o = [torch.full((4, h, w), float('nan'), device='cuda') for _ in range(7)]
    L.check(lib.yond_box_stats_self1_f32(L.ptr(x), H, W, k, k2, tw, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), st), "self1")
    L.check(lib.yond_box_stats_self2_f32(L.ptr(o[2]), h, w, k, tw, L.ptr(o[3]), st), "self2")
    L.check(lib.yond_box_stats_collab_f32(L.ptr(x), L.ptr(xc), H, W, k, tw, L.ptr(o[4]), L.ptr(o[5]), L.ptr(o[6]), st), "collab")
    torch.cuda.synchronize()
    return o


bad = 0
for (H, W, tw, k) in [(256, 256, 0, 29), (200, 328, 0, 29), (62, 70, 0, 13), (700, 1000, 0, 29), (256, 2048, 32, 29), (512, 8192, 256, 29),
                      (130, 902, 0, 29), (64, 64, 0, 5), (3000, 4000, 0, 29)]:
    rng = np.random.default_rng(H * 7 + W)
    x = torch.from_numpy(rng.random((H, W), dtype=np.float32)).cuda()
    xc = torch.from_numpy((rng.random((H, W), dtype=np.float32) * 0.5 + 0.25)).cuda()
    k2 = k // 3 * 2 + 1
    a, b = run(new, x, xc, H, W, k, k2, tw), run(old, x, xc, H, W, k, k2, tw)
    names = ["mean", "var", "blur2", "lap", "c.mean", "c.var", "c.lap"]
    line = []
    for n, p, q in zip(names, a, b):
        assert not torch.isnan(p).any(), (n, "unwritten outputs")
        mu, fr = ulps(p, q)
        line.append("%s %d/%.1e" % (n, mu, fr))
        if mu > 1 or fr > 2e-3:
            bad += 1
    print((H, W, tw, k), " ".join(line), flush=True)
print("max ulp > 1 or > 0.2 % different:", bad)

H, W = 3000, 4000
h, w = H // 2, W // 2
noisy, clean = S.synth_noisy(H, W, 4.0, 6.0, 0)
x = torch.from_numpy(noisy).cuda()
xc = torch.from_numpy(clean).cuda()
o = [torch.empty(4, h, w, device='cuda') for _ in range(4)]


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for nm, lib in (("new", new), ("old", old)):
    print(nm, "self1 %.1f us" % t(lambda: lib.yond_box_stats_self1_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[3]), st)),
          "self2 %.1f us" % t(lambda: lib.yond_box_stats_self2_f32(L.ptr(o[3]), h, w, 29, 0, L.ptr(o[2]), st)),
          "collab %.1f us" % t(lambda: lib.yond_box_stats_collab_f32(L.ptr(x), L.ptr(xc), H, W, 29, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), st)), flush=True)
sys.exit(1 if bad else 0)
