"""Same-process A/B of the SIDD evaluation image (configs[2]: 3000x5328 estimate frame + 32 blocks, `iter`, block metrics) on the device
chain against the host-side chain (pipeline.CHAIN_SIDD): python tools/sidd_chain_ab.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
from yond_public_amd import archs as A, synthetic as S, pipeline as P
dev = torch.device('cuda')
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.denoising_state_dict(net, 0)); net = net.to(dev).eval()
items = Bn.sidd_items(3, dev)
outs = {}
for rep in range(2):
    for flag in (True, False):
        P.CHAIN_SIDD = flag
        for it in items:
            Bn.sidd_eval_item(it, net, arch, P)
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.0:
            res = Bn.sidd_eval_item(items[n % len(items)], net, arch, P)
            n += 1
        torch.cuda.synchronize()
        print("device chain" if flag else "host chain  ", "%.3f ms per image (%d images)" % ((time.perf_counter() - t0) / n * 1e3, n), flush=True)
        outs[flag] = Bn.sidd_eval_item(items[0], net, arch, P)
a, b = outs[True], outs[False]
for r in range(2):
    print("round", r, "max |device - host| =", float((a['raw_dns'][r] - b['raw_dns'][r]).abs().max()), "params", a['params'][r], b['params'][r])
