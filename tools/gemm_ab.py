"""N4: the plain GEMM layers of the training step (1x1 short cuts over two sources, transposed 2x2 layers; forward and backward) on the
split-operand GEMM kernel (train.GEMM_SPLIT) against the fp32-MFMA convolution path, same process: python tools/gemm_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yond_public_amd.train as T
from yond_public_amd import archs as A
dev = torch.device('cuda')
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
torch.manual_seed(0)
net = A.GuidedResUnet(arch); A.initialize_weights(net)
ts = T.TrainStep(net.to(dev), lr=1e-4, ddp=False, graph=False)
P = ts.params
def timeit(f, n=20):
    """Device time per call: n calls captured in one hipGraph and replayed (the host's launch cost stays outside)."""
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 64
for lvl, i in ((0, 9), (1, 8), (2, 7), (3, 6)):
    c = 32 * 2 ** lvl; hw = 128 >> lvl
    up = torch.randn(B, hw, hw, c, device=dev); sk = torch.randn(B, hw, hw, c, device=dev)
    w, b = P[f'conv{i}.short_cut.0.weight'].detach(), P[f'conv{i}.short_cut.0.bias'].detach()
    xin = torch.randn(B, hw // 2, hw // 2, 2 * c, device=dev)
    wt, bt = P[f'upv{i}.weight'].detach(), P[f'upv{i}.bias'].detach()
    res = {}
    with torch.no_grad():
        for flag in (1, 0):
            T.GEMM_SPLIT = bool(flag)
            T.GEMM_MIN_K = 0                                   # (every level on the GEMM kernel: what train.GEMM_MIN_K decides)
            y = T._Conv1x1.apply(up, sk, w, b, ts.plan)
            f1 = timeit(lambda: T._Conv1x1.apply(up, sk, w, b, ts.plan))
            yt = T._ConvT2x2.apply(xin, wt, bt, ts.plan)
            f2 = timeit(lambda: T._ConvT2x2.apply(xin, wt, bt, ts.plan))
            res[flag] = (f1, f2, y.clone(), yt.clone())
    d1 = float((res[1][2] - res[0][2]).abs().max()); d2 = float((res[1][3] - res[0][3]).abs().max())
    print(f"level {lvl} ({c} ch, {hw}x{hw}): 1x1 ({2*c}->{c}) forward {res[1][0]:6.1f} vs {res[0][0]:6.1f} us; "
          f"convT ({2*c}->{c}, from {hw//2}x{hw//2}) forward {res[1][1]:6.1f} vs {res[0][1]:6.1f} us   (split GEMM vs fp32 MFMA, device time; max diff {d1:.1e} {d2:.1e})", flush=True)
