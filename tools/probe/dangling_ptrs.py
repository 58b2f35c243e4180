"""Debug: which device pointers handed to the library during a TrainStep graph capture belong to memory that is FREE in the default
(non-graph) pool afterwards?  (A replay reads / writes them; any later allocation may own that memory.)"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from yond_public_amd import _lib as L, train as TR, archs as A, synthetic as S
rec, on = [], [False]
_ptr = L.ptr
def ptr_dbg(t):
    p = _ptr(t)
    if on[0] and t is not None and hasattr(t, 'data_ptr'):
        fr = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-5:-1]]
        rec.append((t.data_ptr(), t.numel() * t.element_size(), tuple(t.shape), " < ".join(reversed(fr))))
    return p
L.ptr = ptr_dbg
TR.L.ptr = ptr_dbg
from yond_public_amd import engine as E
E.L.ptr = ptr_dbg
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=8, nframes=1, res=True, norm=True)
torch.manual_seed(0)
net = A.GuidedResUnet(dict(arch)).to('cuda')
ts = TR.TrainStep(net, lr=5e-3)
x = torch.rand(4, 4, 32, 32, device='cuda'); y = torch.rand(4, 4, 32, 32, device='cuda'); sg = torch.rand(4, 1, 1, 1, device='cuda') * 0.1 + 0.02
ts.step(x, y, sg); ts.step(x, y, sg)
cap = ts._capture
def cap_dbg(*a, **k):
    on[0] = True
    try:
        return cap(*a, **k)
    finally:
        on[0] = False
ts._capture = cap_dbg
ts.step(x, y, sg)                      # the third step captures
import gc; gc.collect(); torch.cuda.synchronize()
snap = torch.cuda.memory_snapshot()
free_default = []
for seg in snap:
    pool = tuple(seg.get('segment_pool_id', (0, 0)))
    addr = seg['address']
    for b in seg['blocks']:
        if b['state'] != 'active_allocated' and pool == (0, 0):
            free_default.append((addr, addr + b['size']))
        addr += b['size']
seen = set()
for p, nbytes, shape, where in rec:
    for lo, hi in free_default:
        if lo <= p < hi and (p, where) not in seen:
            seen.add((p, where))
            print(f"DANGLING ptr 0x{p:x} {nbytes} B shape {shape}   {where}")
print(f"{len(rec)} pointers recorded during the capture, {len(seen)} dangling")
