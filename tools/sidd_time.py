"""Where the time of one SIDD image goes (cfg 3: [32][256][256] stack, two rounds, block-wise batch-32 forwards)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import archs as A, synthetic as S, pipeline as P
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.denoising_state_dict(net, 0)); net = net.to('cuda').eval()
pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': False}
noisy, clean = S.synth_noisy(256, 8192, 4.0, 6.0, 100)
full, _ = S.synth_noisy(1024, 1536, 4.0, 6.0, 500)
lr_h = np.array(np.split(noisy, 32, axis=-1)); hr_h = np.array(np.split(clean, 32, axis=-1))
lr_d, full_d, hr_d = torch.from_numpy(lr_h).cuda(), torch.from_numpy(full).cuda(), torch.from_numpy(clean).cuda()
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("IterDenoise, host arrays in : %.2f ms" % t(lambda: P.IterDenoise(lr_h, net, arch, pipe, lr_full=full, device='cuda')))
print("IterDenoise, device tensors  : %.2f ms" % t(lambda: P.IterDenoise(lr_d, net, arch, pipe, lr_full=full_d)))
print("  once                       : %.2f ms" % t(lambda: P.IterDenoise(lr_d, net, arch, dict(pipe, iter='once'), lr_full=full_d)))
res = P.IterDenoise(lr_d, net, arch, pipe, lr_full=full_d)
print("block_metrics x2            : %.2f ms" % t(lambda: [P.block_metrics(d, hr_d) for d in res['raw_dns']]))
p = P.default_params(); p['gain'], p['sigma'] = 4.0, 6.0
lut = P.get_bias(np.float32(noisy.max()) * np.float32(959.0), 6.0, 4.0, device='cuda')
print("VST_Denoiser batch 32        : %.2f ms" % t(lambda: P.VST_Denoiser(lr_d, p, net, arch, 'pre', lut, clip01=True)))
print("SimpleNLF self (1024x1536)   : %.2f ms" % t(lambda: P.SimpleNLF(full_d, k=29)))
cat = torch.cat(list(lr_d), dim=-1).contiguous()
print("SimpleNLF collab SIDD_256    : %.2f ms" % t(lambda: P.SimpleNLF(cat, res['raw_dns'][0], k=29, setting={'mode': 'collab', 'SIDD_256': True})))
