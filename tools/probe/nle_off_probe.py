"""What the estimator costs the streamed driver: pipeline.denoise_stream over 96 resident 3000x4000 frames with the estimator + parameter chain queued as usual ("on") and
with their launches skipped after the first frames ("off": the parameter blocks of the ring keep valid contents, the network passes are unchanged).
    python tools/probe/nle_off_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yond_public_amd import archs as A, synthetic as S, pipeline as P
dev = torch.device('cuda')
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.denoising_state_dict(net, 0)); net = net.to(dev).eval()
pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'max_iter': 1, 'full_dn': True}
frames = [torch.from_numpy(S.synth_noisy(3000, 4000, 4.0, 6.5, 10 + i)[0]).to(dev) for i in range(3)]
orig = P._chain_estimate
calls = [0]
def est(*a, **k):
    calls[0] += 1
    if MODE == 'off' and calls[0] > 8:
        return
    return orig(*a, **k)
P._chain_estimate = est
for MODE in ('on', 'off', 'on', 'off'):
    calls[0] = 0
    for _ in P.denoise_stream((frames[i % 3] for i in range(12)), net, arch, pipe): pass
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 96
    for _ in P.denoise_stream((frames[i % 3] for i in range(n)), net, arch, pipe): pass
    torch.cuda.synchronize(); el = time.perf_counter() - t
    print(MODE, round(el / n * 1e3, 3), 'ms per frame', round(n * 12.0 / el, 1), 'MP/s')
