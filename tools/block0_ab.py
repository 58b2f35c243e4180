"""Same-process A/B of one SNR-Net forward at the cfg-2 shape: the two level-0 residual blocks as ONE launch each (csrc/block0_fused.hip,
engine.FUSE_BLOCK0) against the two split-operand launches per block.  Prints the medians of alternating runs and the output difference."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import archs as A, synthetic as S, pipeline as P, engine as E
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.procedural_state_dict(net, 0)); net = net.to('cuda').eval()
plan = P._plan_of(net, torch.device('cuda'))
B, Hh, Ww = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 1504, 2016)
x = torch.rand(B, Hh, Ww, 4, device='cuda'); t = torch.full((B,), 0.03, device='cuda'); ub = x.reshape(B, -1).max(1).values.contiguous()
outs, times = {}, {False: [], True: []}
for rep in range(12):
    for flag in (False, True):
        E.FUSE_BLOCK0 = flag
        for _ in range(2 if rep == 0 else 0):
            plan.forward_nhwc4(x, t, ub=ub)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            y = plan.forward_nhwc4(x, t, ub=ub)
        e1.record(); torch.cuda.synchronize()
        times[flag].append(e0.elapsed_time(e1) / 4 * 1e3)
        outs[flag] = y.clone()
med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
print(f"forward {B}x{Hh}x{Ww}: two launches per level-0 block {med[False]:.1f} us, fused {med[True]:.1f} us ({(med[True] / med[False] - 1) * 100:+.1f} %)")
print("max |difference| of the outputs: %.3e (max |output| %.3f)" % (float((outs[True] - outs[False]).abs().max()), float(outs[False].abs().max())))
