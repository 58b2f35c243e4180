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


def denoising_state_dict(net, seed=0, eps=0.004):
    """Deterministic weights that make the network a (weak but real) denoiser -- out = 3x3 box mean of the
    input plus an eps-sized input-dependent perturbation from all other layers -- so that round 2 of IterDenoise
    (the collaborative estimate, YOND_SIDD.py:419-472) is well posed: with random weights the reference's
    beta1 < 0 guard (:445-447) ends it.  The first layer puts the box mean of input channel c into feature c and
    the input into feature 4+c; every later stage passes them through (second convolutions and every other
    procedural weight scaled by eps; decoder shortcuts = identity on the skip half, centre-tap identities for
    UNetSeeInDark); the output projection takes feature c minus feature 4+c and the network's global residual
    adds the input back."""
    sd = procedural_state_dict(net, seed)
    name = type(net).__name__

    def first_layer(key):
        w, b = sd[key + '.weight'], sd[key + '.bias']
        if w.shape[0] < 8 or w.shape[1] != 4:
            raise ValueError("denoising_state_dict needs nf >= 8 and 4 input channels")
        w[8:] *= eps
        b[8:] *= eps
        w[:8] = 0
        b[:8] = 0
        for c in range(4):
            w[c, c] = 1.0 / 9.0
            w[4 + c, c, 1, 1] = 1.0

    def last_layer(key):
        w, b = sd[key + '.weight'], sd[key + '.bias']
        w *= eps
        b *= 0
        for c in range(4):
            w[c, c, 0, 0] = 1.0
            w[c, 4 + c, 0, 0] = -1.0

    if name == 'UNetSeeInDark':
        first_layer('conv1_1')
        for i in range(1, 10):
            for j in (1, 2):
                if (i, j) == (1, 1):
                    continue
                w, b = sd[f'conv{i}_{j}.weight'], sd[f'conv{i}_{j}.bias']
                w *= eps
                b *= eps
                cout, cin = w.shape[:2]
                skip0 = cin - cout if (i >= 6 and j == 1) else 0          # cat([up, skip]): the skip half comes second
                for c in range(min(cout, cin - skip0)):
                    w[c, skip0 + c, 1, 1] += 1.0
        for i in range(6, 10):
            sd[f'upv{i}.weight'] *= eps
            sd[f'upv{i}.bias'] *= eps
        last_layer('conv10_1')
        return sd

    first_layer('conv_in')
    for i in range(1, 10):
        sd[f'conv{i}.conv2.weight'] *= eps
        sd[f'conv{i}.conv2.bias'] *= eps
        if i >= 6:
            w, b = sd[f'conv{i}.short_cut.0.weight'], sd[f'conv{i}.short_cut.0.bias']
            w *= eps
            b *= eps
            cout = w.shape[0]
            for c in range(cout):
                w[c, cout + c, 0, 0] += 1.0                                # cat([up, skip]): identity on the skip half
    last_layer('conv10')
    return sd
