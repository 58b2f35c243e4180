import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import yond_oracle as O
from yond_public_amd import pipeline as P, archs as A, synthetic as S
torch.set_num_threads(16)
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
H, W = 2048, 3072
noisy, clean = O.synth_noisy(H, W, 4.0, 6.0, 0)
sd = O.denoising_state_dict(arch, 0)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(sd); net = net.to('cuda').eval()
pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'max_iter': 1, 'full_dn': True}
res = P.IterDenoise(torch.from_numpy(noisy).cuda(), net, arch, pipe)
reg = res['regs'][0]
p = O.default_params(); p['gain'], p['sigma'] = reg[0] * 959, np.sqrt(max(reg[1], 0)) * 959
ref, info = O.VST_Denoiser(noisy, p, arch, sd, 'pre', full=True)
ref = ref.clip(0, 1)
got = res['raw_dns'][0].cpu().numpy()
d = np.abs(got - ref)
print("max", d.max(), "n>1e-4", (d > 1e-4).sum(), "argmax", np.unravel_index(d.argmax(), d.shape))
ys, xs = np.nonzero(d > 1e-4)
print("rows", np.unique(ys)[:20], "cols", np.unique(xs)[:20], len(ys))
# K1 alone
x = torch.from_numpy(noisy).cuda()
lut = P.get_bias(np.float32(noisy.max()) * np.float32(959.0), p['sigma'], p['gain'], device=x.device)
lo, hi = P.vst_scalar(0, p['sigma'], p['gain']), P.vst_scalar(959.0, p['sigma'], p['gain'])
from yond_public_amd import _lib as L
lib = L.load()
h, w = H // 2, W // 2
x4 = torch.empty(h, w, 4, device='cuda'); mx = torch.empty(1, device='cuda')
lib.yond_pack_vst_norm_f32(L.ptr(x), H, W, L.ptr(x4), 0, 0, 0, 0, 1, 959.0, float(p['gain']), float(p['sigma']), float(lo), float(hi), L.ptr(lut.x), L.ptr(lut.y), len(lut), L.ptr(mx), L.stream())
k1 = x4.cpu().numpy()
k1ref = info['net_in'][0].permute(1, 2, 0).numpy()
dk = np.abs(k1 - k1ref)
print("K1 max", dk.max(), "argmax", np.unravel_index(dk.argmax(), dk.shape), "n>1e-6", (dk > 1e-6).sum())
i = np.unravel_index(dk.argmax(), dk.shape)
print("K1 at argmax: got", k1[i], "ref", k1ref[i], "input*959", O.bayer2rggb(noisy)[i] * 959)
