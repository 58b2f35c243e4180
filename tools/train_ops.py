"""Which torch operators run inside one eager training step (N4 shape), with their device time: `python tools/train_ops.py`.
The step's own kernels go through ctypes and do not show up here; this lists autograd's glue around them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import archs as A
from yond_public_amd.train import TrainStep
dev = torch.device('cuda')
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
torch.manual_seed(0)
net = A.GuidedResUnet(arch)
A.initialize_weights(net)
ts = TrainStep(net.to(dev), lr=1e-4, ddp=False, graph=False)
g = torch.Generator().manual_seed(1)
hr = torch.rand(64, 4, 128, 128, generator=g).to(dev)
sg = (torch.rand(64, 1, 1, 1, generator=g) * 0.18 + 0.02).to(dev)
lr = (hr + torch.randn(hr.shape, generator=g).to(dev) * sg).clamp(0, 1)
for _ in range(3):
    ts.step(lr, hr, sg)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    ts.step(lr, hr, sg)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
