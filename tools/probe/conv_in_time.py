"""Time the first layer (4 -> 32 channels, 3x3) at the cfg-2 size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yond_public_amd import archs as A, synthetic as S, pipeline as P, _lib as L
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.procedural_state_dict(net, 0)); net = net.to('cuda').eval()
plan = P._plan_of(net, torch.device('cuda'))
lib = L.load()
N, H, W = 1, 1504, 2016
x4 = torch.rand(N, H, W, 4, device='cuda'); ub = x4.reshape(N, -1).max(1).values.contiguous()
a = torch.empty(N, H, W, 32, device='cuda')
st = L.stream()
fn = lambda: L.check(lib.yond_conv_in_f32(L.ptr(x4), L.ptr(ub), N, H, W, 32, L.ptr(plan.conv_in_w), L.ptr(plan.conv_in_b), 0.01, L.ptr(a), st), "conv_in")
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print("conv_in: %.1f us, checksum %.6f" % (e0.elapsed_time(e1) / 20 * 1e3, float(a.double().sum())))
