"""Synthetic inputs and procedural weights (there is no network for datasets or checkpoints):
seeded Poisson-Gaussian Bayer frames and a deterministic weight filler for the
three hot-path architectures.  Must stay bit-identical to the generators the golden fixtures were
made with (tests/test_synthetic.py checks that)."""
import math
import zlib

import numpy as np
import torch


def synth_clean(H, W):
    """Smooth ramp + 256-px checker + mild sinusoid in [0,1]: flat regions and edges."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    ramp = 0.08 + 0.55 * (x / max(W - 1, 1)) * (0.6 + 0.4 * y / max(H - 1, 1))
    checker = 0.12 * ((((x // 256) + (y // 256)) % 2) - 0.5)
    wave = 0.004 * np.sin(2 * np.pi * x / 97.0) * np.cos(2 * np.pi * y / 131.0)
    return np.clip(ramp + checker + wave, 0.0, 1.0)


def synth_noisy(H, W, K=4.0, sigma=6.0, idx=0, clip=True, scale=959.0):
    rng = np.random.default_rng(1997 + idx)
    clean = synth_clean(H, W)
    noisy = (rng.poisson(clean * scale / K) * K + rng.normal(0.0, sigma, (H, W))) / scale
    if clip:
        noisy = np.clip(noisy, 0, 1)
    return noisy.astype(np.float32), clean.astype(np.float32)


def procedural_state_dict(net, seed=0, gain=1.0):
    """Deterministic weights for a yond_public_amd.archs module: per-key seeded generator,
    fan-in-scaled normal weights, small biases, FiLM scales around 1."""
    sd = {}
    for key, ref in sorted(net.state_dict().items()):
        shape = tuple(ref.shape)
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
        if key.endswith('.weight'):
            fan_in = shape[0] if 'upv' in key else shape[1] * shape[2] * shape[3]
            std = gain * math.sqrt(1.0 / fan_in)
            if '.gamma.0' in key or '.sfm1.0' in key or '.sfm2.0' in key:
                std = 4.0
            sd[key] = torch.randn(shape, generator=g) * std
        else:
            sd[key] = torch.randn(shape, generator=g) * 0.05
            if '.gamma.2' in key or '.sfm1.2' in key or '.sfm2.2' in key:
                sd[key] = sd[key] + 1.0
    return sd
