"""The launch sequence of ONE steady-state frame of the default path (for `rocprofv3 --kernel-trace`): 6 frames through
denoise_stream; tools/frame_trace_summary.py prints what runs between two K1 launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import archs as A, synthetic as S, pipeline as P
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.denoising_state_dict(net, 0)); net = net.to('cuda').eval()
pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'max_iter': 1, 'full_dn': True, 'collab_sidd256': False}
frames = [torch.from_numpy(S.synth_noisy(3000, 4000, 4.0, 6.0, i)[0]).cuda() for i in range(2)]
for _ in P.denoise_stream((frames[i % 2] for i in range(6)), net, arch, pipe):
    pass
torch.cuda.synchronize()
