"""N4 measurement: training steps per second at the reference's training shape (runfiles/Gaussian/GRU_5to50_norm_mix.yml: GuidedResUnet
nf 32, batch 64 of 256 x 256 Bayer patches = [64][4][128][128]), on the HIP training path (yond_public_amd/train.py).
    python tools/train_bench.py [--net GuidedResUnet|UNetSeeInDark] [--batch 64] [--steps 10]
Prints one JSON line: ms per step, patches / s, Bayer MP / s, the nominal forward + backward convolution FLOPs and TF/s."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from yond_public_amd import archs as A
from yond_public_amd.train import TrainStep


def conv_flops(net, N, h, w):
    """2 * Cout * Cin * k * k * output pixels per convolution, every layer at its own resolution (forward); x3 for a step."""
    total = 0
    for name, m in net.named_modules():
        if isinstance(m, torch.nn.Conv2d) and m.in_channels > 1:
            lvl = {32: 0, 64: 1, 128: 2, 256: 3, 512: 4}
            nf = net.args['nf']
            r = min(m.out_channels, m.in_channels if m.in_channels >= nf else m.out_channels) // nf
            s = max(r.bit_length() - 1, 0)
            px = N * (h >> s) * (w >> s)
            total += 2 * m.out_channels * m.in_channels * m.kernel_size[0] ** 2 * px
        elif isinstance(m, torch.nn.ConvTranspose2d):
            s = (m.out_channels // net.args['nf']).bit_length() - 1
            total += 2 * m.out_channels * m.in_channels * N * (h >> s) * (w >> s)
    return total


ap = argparse.ArgumentParser()
ap.add_argument('--net', default='GuidedResUnet')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--nf', type=int, default=32)
ap.add_argument('--conv', default='split')
ap.add_argument('--graph', type=int, default=1)
ap.add_argument('--set', action='append', default=[], help='A/B: NAME=VALUE module attributes of yond_public_amd.train (e.g. WGRAD_BIAS=0)')
a = ap.parse_args()
import yond_public_amd.train as _T
for kv in a.set:
    k, v = kv.split('=')
    setattr(_T, k, type(getattr(_T, k))(int(v)))
dev = torch.device('cuda', 0)
arch = dict(name=a.net, in_nc=4, out_nc=4, nf=a.nf, nframes=1, res=True, norm=True)
if a.net != 'UNetSeeInDark':
    arch['guided'] = True
torch.manual_seed(0)
net = getattr(A, a.net)(arch)
A.initialize_weights(net)
net = net.to(dev)
ts = TrainStep(net, lr=1e-4, ddp=False, conv=a.conv, graph=bool(a.graph))
g = torch.Generator(device='cpu').manual_seed(1)
hr = torch.rand(a.batch, 4, 128, 128, generator=g).to(dev)
sigma = (torch.rand(a.batch, 1, 1, 1, generator=g) * 0.18 + 0.02).to(dev)
lr = (hr + torch.randn(hr.shape, generator=g).to(dev) * sigma).clamp(0, 1)
sg = sigma if 'guided' in arch else None
for _ in range(4):                                           # (two eager steps, the capture, one replay)
    ts.step(lr, hr, sg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss, _ = ts.step(lr, hr, sg)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.steps * 1e3
fl = conv_flops(net, a.batch, 128, 128)
print(json.dumps({"net": a.net, "nf": a.nf, "conv": a.conv, "batch": a.batch, "ms_per_step": round(ms, 3), "patches_per_s": round(a.batch / ms * 1e3, 1),
                  "bayer_mp_per_s": round(a.batch * 256 * 256 / ms / 1e3, 1), "fwd_conv_gflop": round(fl / 1e9, 1),
                  "step_tflops_nominal": round(3 * fl / ms / 1e9, 1), "loss": loss, "loss_scale": getattr(ts, 'last_scale', None)}))
