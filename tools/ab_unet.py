"""Same-process A/B of UNetSeeInDark's forward (cfg 4 network, one 1504 x 2016 image): stage-internal tensors [N][H][W][C] vs split planes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import archs as A, synthetic as S, pipeline as P, engine as E
arch = dict(name='UNetSeeInDark', in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.UNetSeeInDark(dict(arch)); net.load_state_dict(S.procedural_state_dict(net, 0)); net = net.to('cuda').eval()
plan = P._plan_of(net, torch.device('cuda'))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.rand(N, 1504, 2016, 4, device='cuda'); ub = x.reshape(N, -1).max(1).values.contiguous()
res, outs = {}, {}
for rep in range(7):
    for name, flag in (('nhwc', False), ('split planes', True)):
        E.UNET_SP = flag
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = plan.forward_nhwc4(x, None, ub=ub)
        e1.record(); torch.cuda.synchronize()
        if rep >= 2: res.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3 / N)
        outs[name] = y
for k, v in res.items():
    v = sorted(v); print("%-14s forward median %.1f us per image" % (k, v[len(v) // 2]))
a, b = outs.values()
print("max |difference of the outputs| = %.3e" % float((a - b).abs().max()))
