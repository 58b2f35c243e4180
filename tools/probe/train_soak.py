"""Soak of the captured training step under poisoned allocations: idle gaps, allocate-and-drop between steps, an engine forward now and then;
every step's gradients must stay finite and small and equal those of a second TrainStep WITHOUT graphs that starts the step from the same state
(to 20x what two eager steps differ by: float atomics).
    python tools/probe/train_soak.py [steps] [nf]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
_empty, _empty_like = torch.empty, torch.empty_like
def _poison(t):
    if t.is_floating_point() and t.device.type == 'cuda' and t.numel():
        t.fill_(float('nan'))
    return t
torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
from yond_public_amd import train as TR, archs as A
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=nf, nframes=1, res=True, norm=True)
torch.manual_seed(0)
net_a = A.GuidedResUnet(dict(arch)).to('cuda')
net_b = A.GuidedResUnet(dict(arch)).to('cuda'); net_b.load_state_dict(net_a.state_dict())
ts_a = TR.TrainStep(net_a, lr=1e-3, graph=True, charbonnier=True)      # (a smooth loss: L1 gradients jump by 1 / n where pred crosses the target)
ts_b = TR.TrainStep(net_b, lr=1e-3, graph=False, charbonnier=True)
net_c = A.GuidedResUnet(dict(arch)).to('cuda'); net_c.load_state_dict(net_a.state_dict())
ts_c = TR.TrainStep(net_c, lr=1e-3, graph=False, charbonnier=True)                # a second eager step: how much two runs of the same step differ (atomic sums)
noise = 0.0
g = torch.Generator(device='cuda').manual_seed(1)
worst = 0.0
for i in range(steps):
    x = torch.rand(4, 4, 32, 32, device='cuda', generator=g); y = (x * 0.9).contiguous()
    sg = torch.rand(4, 1, 1, 1, device='cuda', generator=g) * 0.1 + 0.02
    la, ga = ts_a.step(x, y, sg)
    lb, gb = ts_b.step(x, y, sg)
    m = max(float(v.abs().max()) for v in ga.values())
    worst = max(worst, m)
    lc, gc = ts_c.step(x, y, sg)
    d = max(float((ga[k] - gb[k]).abs().max()) for k in ga)
    dn = max(float((gc[k] - gb[k]).abs().max()) for k in ga)
    noise = max(noise, dn)
    if not (m < 1e3) or d > 20 * max(noise, 1e-7):
        kk = max(ga, key=lambda k: float((ga[k] - gb[k]).abs().max()))
        print(f"step {i}: graph-step gradients differ from the eager step's: max |g| {m:g}, max difference {d:g} in {kk} (eager vs eager so far: {noise:g}), losses {la} {lb}")
        sys.exit(1)
    # the three nets continue from the SAME state (the graph step's): Adam turns a 1e-8 difference in a tiny gradient into a 1e-5 difference
    # of a parameter within tens of steps, and the comparison would measure that drift instead of one step
    for other in (ts_b, ts_c):
        other.arena.copy_(ts_a.arena); other.adam_m.copy_(ts_a.adam_m); other.adam_v.copy_(ts_a.adam_v)
        other.plan.wbatch = None
        other.m._plan = None
    if i % 7 == 3:
        time.sleep(0.03)                                         # an idle GPU before the next replay
    if i % 11 == 5:
        for sz in (256, 4096, 65536, 1 << 20, 1 << 22):
            xs = [torch.empty(sz, device='cuda') for _ in range(4)]
            del xs
    if i % 25 == 12:
        with torch.no_grad():
            net_a(x, sg)                                         # the engine's forward (another plan, other buffers)
pa = torch.cat([p.reshape(-1) for p in net_a.parameters()]); pb = torch.cat([p.reshape(-1) for p in net_b.parameters()])
print(f"eager vs eager gradient noise (atomic sums): {noise:.3g}")
print(f"{steps} steps, nf {nf}: every graph step's gradients equal the eager step's from the same state (largest |gradient| {worst:.3g}, max |p| {float(pa.abs().max()):.3g})")
