"""Same-process A/B of the data-flow variants of one SNR-Net forward at the cfg-2 shape: per-launch medians (HIP events)
for  nhwc: every tensor [N][H][W][C] float32;  tmp: block-internal tensors in split planes;  flow: the whole forward in the
split-plane data flow.  Configurations are interleaved round by round (cdna_hip_programming.md rule 24)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import archs as A, synthetic as S, pipeline as P, engine as E
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.procedural_state_dict(net, 0)); net = net.to('cuda').eval()
plan = P._plan_of(net, torch.device('cuda'))
x = torch.rand(1, 1504, 2016, 4, device='cuda'); t = torch.full((1,), 0.03, device='cuda'); ub = x.reshape(1, -1).max(1).values.contiguous()
CFG = {'nhwc': (False, False, False, False, 99), 'tmp': (True, False, False, False, 99), 'flow': (True, True, False, False, 99), 'snake': (True, True, True, False, 99),
       'sub2': (True, True, True, True, 99), 'c1dma3': (True, True, True, True, 3), 'c1dma2': (True, True, True, True, 2)}   # snake: alternating tile order; sub2: + two sub-positions per tile in the 1->0 decoder GEMM
names = sys.argv[1:] or list(CFG)
ref = None
acc = {n: {} for n in names}
tags = {}
for rep in range(9):
    for n in names:
        E.SPLIT_PLANES, E.SP_FLOW, E.SNAKE_ORDER, E.K1_SUB2, E.SP_CONV1_MIN_LEVEL = CFG[n]
        plan.prof = [] if rep >= 2 else None
        y = plan.forward_nhwc4(x, t, ub=ub)
        torch.cuda.synchronize()
        if rep == 0:
            ref = y.clone() if ref is None else ref
            print(n, "max |out - first config| =", float((y - ref).abs().max()))
        if plan.prof:
            for i, (tag, fl, e0, e1) in enumerate(plan.prof):
                acc[n].setdefault(i, []).append(e0.elapsed_time(e1) * 1e3)
                tags.setdefault(i, tag)
        plan.prof = None
keys = sorted(acc[names[0]])
tot = {n: 0.0 for n in names}
print("%-34s" % "launch" + "".join("%10s" % n for n in names))
for k in keys:
    row = []
    for n in names:
        v = sorted(acc[n][k]); m = v[len(v) // 2]; tot[n] += m; row.append(m)
    print("%2d %-31s" % (k, tags[k]) + "".join("%10.1f" % m for m in row))
print("%-34s" % "sum of conv launches (us)" + "".join("%10.1f" % tot[n] for n in names))
