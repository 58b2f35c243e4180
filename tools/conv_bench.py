"""Per-layer timing of the convolution kernels at the cfg-2 shapes (3000x4000 Bayer -> 1504x2016 packed).
Usage: python tools/conv_bench.py [--reps 5] [--only 3x3s1]   -- prints one line per layer."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import _lib as L
from yond_public_amd.engine import _PackedConv, DenoiserPlan

DEV = 'cuda:0'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--H", type=int, default=1504)
    ap.add_argument("--W", type=int, default=2016)
    ap.add_argument("--algo", default="", help="'' (engine default), 0, 1, 2, split, half")
    a = ap.parse_args()
    plan = DenoiserPlan.__new__(DenoiserPlan)
    plan.lib, plan.dev, plan.prof = L.load(), torch.device(DEV), None
    H0, W0 = a.H, a.W
    layers = []
    for lvl, C in enumerate([32, 64, 128, 256, 512]):
        h, w = H0 >> lvl, W0 >> lvl
        layers.append((f"3x3s1 C{C} L{lvl} silu+film", 3, 1, [C], C, h, w, dict(pre_act=1, film=True), False))
        layers.append((f"3x3s1 C{C} L{lvl} silu+res", 3, 1, [C], C, h, w, dict(pre_act=1, res=True), False))
        if lvl < 4:
            layers.append((f"3x3s2 C{C}->{2*C} L{lvl}", 3, 2, [C], 2 * C, h, w, {}, False))
            layers.append((f"convT C{2*C}->{C} L{lvl+1}->L{lvl}", 1, 1, [2 * C], C, h // 2, w // 2, {}, True))
            layers.append((f"1x1 2x{C}->{C} L{lvl}", 1, 1, [C, C], C, h, w, {}, False))
    g = torch.Generator().manual_seed(0)
    for name, ks, st, splits, cout, h, w, opts, shuffle in layers:
        if a.only and a.only not in name:
            continue
        cin = sum(splits)
        if shuffle:
            wt = torch.randn(cin, cout, 2, 2, generator=g) / cin ** 0.5
        else:
            wt = torch.randn(cout, cin, ks, ks, generator=g) / (ks * cin ** 0.5)
        pc = _PackedConv(torch.device(DEV), wt, torch.randn(cout, generator=g), ks, st, splits, shuffle=shuffle)
        xs = [torch.randn(1, h, w, c, device=DEV) for c in splits]
        if shuffle:
            dst = torch.empty(1, 2 * h, 2 * w, cout, device=DEV)
        elif st == 2:
            dst = torch.empty(1, h // 2, w // 2, cout, device=DEV)
        else:
            dst = torch.empty(1, h, w, cout, device=DEV)
        kw = {}
        if opts.get('film'):
            kw.update(escale=torch.randn(1, cout, device=DEV), eshift=torch.randn(1, cout, device=DEV), ebatch=1)
        if opts.get('res'):
            kw.update(res=torch.randn_like(dst))
        kw.update({k: v for k, v in opts.items() if k in ('pre_act', 'post_act')})
        if a.algo:
            kw['algo'] = int(a.algo) if a.algo.isdigit() else a.algo
        run = lambda: plan._conv(pc, xs[0], xs[1] if len(xs) > 1 else None, 1, h, w, dst, **kw)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        mpix = (h // st) * (w // st) if not shuffle else h * w
        fl = 2.0 * pc.macs_per_pixel * mpix
        print(f"{name:34s} {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  ({fl/1e9:6.2f} GF)", flush=True)


if __name__ == "__main__":
    main()
