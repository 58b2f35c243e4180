"""Where one eager training step (N4 shape) calls torch for data movement: every call of contiguous() on a non-contiguous tensor, copy_, clone, cat, pad,
zeros / zeros_like / new_zeros, fill_ / zero_, add / add_, mul_, gather / index_select is logged with its call site in yond_public_amd/ and the bytes it moves.
    python tools/probe/train_glue_sites.py"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from yond_public_amd import archs as A
from yond_public_amd.train import TrainStep
LOG = collections.OrderedDict()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'yond_public_amd' in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:100]}"
    return "(outside the package: autograd engine / torch internals)"


def note(op, nbytes):
    if ON[0]:
        k = (op, site())
        a = LOG.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += int(nbytes)


def wrap_method(name, cond=lambda self, *a, **k: True, size=lambda self, *a, **k: self.numel() * self.element_size()):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        if self.is_cuda and cond(self, *a, **k):
            note(name, size(self, *a, **k))
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)


wrap_method('contiguous', cond=lambda self, *a, **k: not self.is_contiguous())
for n in ('copy_', 'clone', 'fill_', 'zero_', 'add', 'add_', 'mul_', 'mul', 'gather', 'index_select', 'new_zeros', '__getitem__'):
    if n == '__getitem__':
        continue
    wrap_method(n)
for mod, name in ((torch, 'cat'), (torch, 'zeros'), (torch, 'zeros_like'), (F, 'pad'), (F, 'leaky_relu'), (F, 'silu'), (torch, 'gather'), (torch, 'index_select')):
    orig = getattr(mod, name)

    def g(*a, _o=orig, _n=name, **k):
        r = _o(*a, **k)
        if isinstance(r, torch.Tensor) and r.is_cuda:
            note(_n, r.numel() * r.element_size())
        return r
    setattr(mod, name, g)

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
ON[0] = True
ts.step(lr, hr, sg)
torch.cuda.synchronize()
ON[0] = False
for (op, where), (n, b) in sorted(LOG.items(), key=lambda kv: -kv[1][1]):
    if b >= 1 << 20:
        print(f"{b / 1e6:9.1f} MB {n:3d} x {op:12s} {where}")
print("(entries below 1 MB omitted:", sum(1 for v in LOG.values() if v[1] < 1 << 20), "sites)")
