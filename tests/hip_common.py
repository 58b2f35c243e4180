"""Shared helpers for the -m gpu parity tests (HIP path vs oracle / golden fixtures)."""
import hashlib

import numpy as np
import torch

ARCHS = {
    "gru32": dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
    "gru8": dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=8, nframes=1, res=True, norm=True),
    "gru32_nonorm": dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=False, norm=False),
    "snr32": dict(name='SNRnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
    "snr8": dict(name='SNRnet', guided=True, in_nc=4, out_nc=4, nf=8, nframes=1, res=True, norm=True),
    "unet32": dict(name='UNetSeeInDark', in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
    "unet8": dict(name='UNetSeeInDark', in_nc=4, out_nc=4, nf=8, nframes=1, res=True, norm=True),
}


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def make_net(arch, seed, device='cuda:0'):
    import yond_oracle as O
    from yond_public_amd import archs as A
    sd = O.procedural_state_dict(arch, seed)
    net = getattr(A, arch['name'])(dict(arch))
    net.load_state_dict(sd)
    return net.to(device).eval(), sd


def report(name, got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    print(f"[parity] {name}: max_abs={err.max():.3e} mean_abs={err.mean():.3e} ref_absmax={np.abs(ref).max():.3e}")
    return err.max()
