"""Function seam of utils/isp_ops.py:57-71 -- Bayer <-> packed RGGB, bit-exact index permutations run
as HIP copy kernels.  NumPy in -> NumPy out (as the reference), device tensor in -> device tensor out."""
import numpy as np
import torch

from .. import pipeline as _P


def _wrap(fn, x):
    if isinstance(x, np.ndarray):
        return fn(x).cpu().numpy()
    return fn(x)


def bayer2rggb(bayer):
    """utils/isp_ops.py:57-59: (H, W) -> (H/2, W/2, 4), channel = 2*dy + dx."""
    return _wrap(_P.bayer2rggb, bayer)


def rggb2bayer(rggb):
    """utils/isp_ops.py:61-63."""
    return _wrap(_P.rggb2bayer, rggb)


def bayer2rggbs(bayers):
    """utils/isp_ops.py:65-67 (batched torch version): (..., H, W) -> (-1, H/2, W/2, 4)."""
    H, W = bayers.shape[-2:]
    flat = bayers.reshape(-1, H, W)
    return torch.stack([_P.bayer2rggb(b) for b in flat])


def rggb2bayers(rggbs):
    """utils/isp_ops.py:69-71."""
    H, W, _ = rggbs.shape[-3:]
    flat = rggbs.reshape(-1, H, W, 4)
    return torch.stack([_P.rggb2bayer(r) for r in flat])
