"""Soak of the two-stream driver (pipeline.denoise_stream: the estimator of frame k + 1 on a side stream under the network of frame k) under
poisoned allocations: many frames of several contents and TWO sizes, idle gaps and foreign allocations between frames; every result must equal
what IterDenoise gives for that frame alone (to the 5e-6 two IterDenoise runs differ by: float64 atomics in the estimator's moment sums).
    python tools/probe/stream_soak.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
_empty, _empty_like = torch.empty, torch.empty_like
def _poison(t):
    if t.is_floating_point() and t.device.type == 'cuda' and t.numel():
        t.fill_(float('nan'))
    return t
torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
import yond_public_amd.pipeline as P
import yond_public_amd.archs as A
import yond_public_amd.synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda:0')
arch = dict(name='GuidedResUnet', in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True, guided=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.procedural_state_dict(net, 0)); net = net.to(dev).eval()
pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': os.environ.get('MODE', 'once'), 'max_iter': 1, 'full_dn': True}     # MODE=iter: the shipped two-round mode
if os.environ.get('WEIGHTS') == 'denoise':
    net.load_state_dict(S.denoising_state_dict(net, 0)); net = net.to(dev).eval()        # (round 2 then runs; with procedural weights it ends at the beta1 < 0 guard)
kinds = [torch.from_numpy(S.synth_noisy(h, w, 3.0 + i, 5.0 + 2 * i, 10 + i)[0]).to(dev) for i, (h, w) in enumerate([(256, 320), (512, 768), (256, 320), (384, 512), (512, 768)])]
ref = [P.IterDenoise(f, net, arch, pipe) for f in kinds]
torch.cuda.synchronize()
order = [(7 * i + i // 5) % len(kinds) for i in range(n)]
def gen():
    for i, k in enumerate(order):
        if i % 6 == 2:
            time.sleep(0.02)
        if i % 9 == 4:
            xs = [torch.empty(sz, device=dev) for sz in (4096, 1 << 18, 1 << 22, 1 << 24)]
            del xs
        yield kinds[k]
worst = 0.0
for i, r in enumerate(P.denoise_stream(gen(), net, arch, pipe)):
    k = order[i]
    assert len(r['raw_dns']) == len(ref[k]['raw_dns'])
    d = max(float((a_ - b_).abs().max()) for a_, b_ in zip(r['raw_dns'], ref[k]['raw_dns']))
    dr = float(np.abs(np.asarray(r['regs'], np.float64) / np.asarray(ref[k]['regs'], np.float64) - 1.0).max())
    worst = max(worst, d)
    if not (d <= 5e-6 and dr <= (1e-9 if pipe['iter'] == 'once' else 3e-4)):
        print(f"frame {i} (content {k}): output differs from IterDenoise's by {d:g}, estimate by {dr:g} (relative)")
        sys.exit(1)
print(f"{n} frames through denoise_stream: every one equal to its IterDenoise result (largest difference {worst:.3g}; max |output| {max(float(q['raw_dns'][0].abs().max()) for q in ref):.3g}, dtype {ref[0]['raw_dns'][0].dtype})")
