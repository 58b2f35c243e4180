"""Same-process A/B of two builds of libyond_hip.so on one SNR-Net forward at the cfg-2 shape: per-launch HIP-event times of this
build against another library (default tools/probe/libyond_hip_r3.so, round 3's), interleaved, plus the largest output difference.
    python tools/ab_lib.py [other.so] [--unet] [--fp16]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import _lib as L, archs as A, synthetic as S, pipeline as P, engine as E

args = [a for a in sys.argv[1:] if not a.startswith("--")]
other = args[0] if args else os.path.join(os.path.dirname(__file__), "probe", "libyond_hip_r3.so")
new = L.load()
old = C.CDLL(other)
for name, a in L.PROTOTYPES.items():
    if hasattr(old, name):
        f = getattr(old, name)
        f.argtypes, f.restype = a, (C.c_size_t if name in L._SIZE_T_RET else C.c_int)
unet = "--unet" in sys.argv
if unet:
    arch = dict(name='UNetSeeInDark', in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
    mk = A.UNetSeeInDark
else:
    arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
    mk = A.GuidedResUnet
net = mk(dict(arch)); net.load_state_dict(S.procedural_state_dict(net, 0)); net = net.to('cuda').eval()
if "--fp16" in sys.argv:                             # the BASELINE cfg 5 path at its own shape
    net.precision = 'fp16'
HW = (2016, 3008) if "--fp16" in sys.argv else (1504, 2016)
dev = torch.device('cuda')
plans = {"new": E.DenoiserPlan(net, dev), "old": E.DenoiserPlan(net, dev)}
plans["old"].lib = old
torch.manual_seed(0)
x = torch.rand(1, HW[0], HW[1], 4, device='cuda'); t = torch.full((1,), 0.03, device='cuda'); ub = x.reshape(1, -1).max(1).values.contiguous()
outs = {}
for k, p in plans.items():
    for _ in range(3):
        outs[k] = p.forward_nhwc4(x, t if not unet else None, ub=ub).clone()
torch.cuda.synchronize()
print("max |new - old| = %.3e  (max |out| %.3f)" % ((outs["new"] - outs["old"]).abs().max().item(), outs["old"].abs().max().item()))
acc = {k: {} for k in plans}
for rep in range(7):
    for k, p in plans.items():
        p.prof = []
        p.forward_nhwc4(x, t if not unet else None, ub=ub)
        torch.cuda.synchronize()
        for i, (tag, fl, e0, e1) in enumerate(p.prof):
            acc[k].setdefault((i, tag), []).append(e0.elapsed_time(e1) * 1e3)
        p.prof = None
tn = to = 0
for key in sorted(acc["new"]):
    a = sorted(acc["new"][key])[len(acc["new"][key]) // 2]
    b = sorted(acc["old"][key])[len(acc["old"][key]) // 2]
    tn += a; to += b
    print(f"{key[0]:2d} {key[1]:32s} new {a:8.1f} us   old {b:8.1f} us   {a - b:+7.1f}")
print("sum of conv launches: new %.1f us   old %.1f us   %+.1f (%.1f %%)" % (tn, to, tn - to, (tn / to - 1) * 100))
