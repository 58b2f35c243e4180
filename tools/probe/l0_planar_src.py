"""Probe: a level-0 3x3 layer (32 -> 32 channels, 1504x2016) reading its input as ONE NHWC tensor (128 B per pixel, of which a
16-channel step takes 64) against TWO 16-channel tensors (every step reads whole 64-byte pixels) through the kernel's
two-source path.  Same weights, same result; only the addresses of the staged loads differ."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yond_public_amd import archs as A, synthetic as S, pipeline as P
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.procedural_state_dict(net, 0)); net = net.to('cuda').eval()
plan = P._plan_of(net, torch.device('cuda'))
N, h, w = 1, 1504, 2016
pc = plan.blocks[1]['conv1']
x = torch.randn(N, h, w, 32, device='cuda')
x0, x1 = x[..., :16].contiguous(), x[..., 16:].contiguous()
res = torch.randn(N, h, w, 32, device='cuda')
o_a, o_b = torch.empty_like(x), torch.empty_like(x)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


keep = pc.psplits
for r, name in ((None, "conv1 (no residual)"), (res, "conv2 (residual)")):
    pc.psplits = keep
    ta = t(lambda: plan._conv(pc, x, None, N, h, w, o_a, res=r, pre_act=1))
    pc.psplits = (16, 16)
    tb = t(lambda: plan._conv(pc, x0, x1, N, h, w, o_b, res=r, pre_act=1))
    pc.psplits = keep
    print(f"{name}: one NHWC32 source {ta:.1f} us, two NHWC16 sources {tb:.1f} us, equal {torch.equal(o_a, o_b)}")
