"""Function seam of utils/isp_algos.py for the hot path: VST, inverse_VST, get_bias, stdfilt, varfilt,
polyfit -- same names and argument meaning, executed by the HIP kernels (no CPU compute path).
NumPy arrays are uploaded and the result comes back as NumPy (the reference's convention); device
tensors stay on the device.  Scalars (the reference calls VST(0, ...) / VST(scale, ...)) are evaluated on
the host in float64."""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L
from .. import pipeline as _P
from ..pipeline import get_bias, DeviceBiasLUT, BiasLUT  # noqa: F401  (utils/isp_algos.py:98-140, 162-231)


def _is_scalar(x):
    return np.isscalar(x) or (isinstance(x, np.ndarray) and x.ndim == 0) or (torch.is_tensor(x) and x.ndim == 0 and not x.is_cuda)


def VST(x, sigma, mu=0, gain=1.0):
    """utils/isp_algos.py:5-14: 2/gain * sqrt(max(gain*x + 3/8 gain^2 + sigma^2 - gain*mu, 0))."""
    if _is_scalar(x):
        fz = np.float64(gain) * float(x) + (3 / 8) * np.float64(gain) ** 2 + np.float64(sigma) ** 2 - np.float64(gain) * mu
        return 2 / np.float64(gain) * np.maximum(fz, 0) ** 0.5
    back = isinstance(x, np.ndarray)
    xd = _P._dev(x)
    out = torch.empty(xd.shape, dtype=torch.float64, device=xd.device)
    L.check(L.load().yond_vst_elem_f32(L.ptr(xd), xd.numel(), float(sigma), float(mu), float(gain), L.ptr(out), L.stream()),
            "yond_vst_elem_f32")
    return out.cpu().numpy() if back else out


def inverse_VST(z, sigma, gain=1, exact=False):
    """utils/isp_algos.py:17-33 (algebraic inverse, or the closed-form exact unbiased inverse).  Unlike the
    reference's exact branch it never modifies `z` in place."""
    if _is_scalar(z):
        z = np.float64(z)
        sg = np.float64(sigma) / np.float64(gain)
        if exact:
            fz = 0.0 if z <= 0 else (z / 2) ** 2 + 0.25 * 1.5 ** 0.5 / z - 1.375 / z ** 2 + 0.625 * 1.5 ** 0.5 / z ** 3 - 0.125 - sg ** 2
        else:
            fz = (z / 2) ** 2 - 0.375 - sg ** 2
        return max(fz, 0.0) * np.float64(gain)
    back = isinstance(z, np.ndarray)
    if back:
        zd = torch.from_numpy(np.ascontiguousarray(z, dtype=np.float64)).to('cuda')
    else:
        if not z.is_cuda:
            raise L.YondHipError("inverse_VST needs a device tensor or a NumPy array (no CPU path)")
        zd = z.contiguous().double()
    out = torch.empty_like(zd)
    L.check(L.load().yond_ivst_elem_f64(L.ptr(zd), zd.numel(), float(sigma), float(gain), int(bool(exact)), L.ptr(out), L.stream()),
            "yond_ivst_elem_f64")
    return out.cpu().numpy() if back else out


def _planes(img):
    """(h, w) or (h, w, C) -> device planar (C, h, w) in groups of 4 channels for the box kernels."""
    back = isinstance(img, np.ndarray)
    t = _P._dev(img)
    if t.dim() == 2:
        t = t[:, :, None]
    return t.permute(2, 0, 1).contiguous(), back


def stdfilt(img, k=5):
    """utils/isp_algos.py:234-242: sqrt(max(blur(img^2) - blur(img)^2, 0)), cv2.blur semantics, per channel."""
    lib = L.load()
    pl, back = _planes(img)
    Cn, h, w = pl.shape
    pad = (-Cn) % 4
    if pad:                                   # the kernel's grid covers 4 planes: fill up with copies of the last one
        pl = torch.cat([pl, pl[-1:].expand(pad, h, w)], 0).contiguous()
    out = torch.empty_like(pl)
    for c0 in range(0, pl.shape[0], 4):       # the kernel filters 4 planes per launch
        L.check(lib.yond_box_stats_self2_f32(L.ptr(pl[c0:c0 + 4]), h, w, int(k), 0, L.ptr(out[c0:c0 + 4]), L.stream()),
                "yond_box_stats_self2_f32")
    res = out[:Cn].permute(1, 2, 0)
    if np.ndim(img) == 2 or (torch.is_tensor(img) and img.dim() == 2):
        res = res[:, :, 0]
    res = res.contiguous()
    return res.cpu().numpy() if back else res


def varfilt(img, k=5):
    """utils/isp_algos.py:245-253 up to the final max(.,0): returned as stdfilt(img, k)**2 (>= 0)."""
    s = stdfilt(img, k)
    return s * s


def polyfit(x, y, ransac=False, clip=False):
    """utils/isp_algos.py:345-365, least-squares branch: non-saturation mask 1e-4 < x < 0.8 when it keeps more than
    1 % of the points, then the line fit -- from five moment sums accumulated on the device."""
    if ransac:
        raise NotImplementedError("RANSAC fit (unused by YOND_SIDD.py:86)")
    xd, yd = _P._dev(x).reshape(-1), _P._dev(y).reshape(-1)
    lap = torch.zeros(xd.numel(), dtype=torch.float32, device=xd.device)          # every point is below th = inf
    th = torch.full((1,), float('inf'), dtype=torch.float64, device=xd.device)
    m = _P._moments(lap, xd.contiguous(), yd.contiguous(), th).cpu().numpy()
    return _P._fit_from_moments(m[0], m[1])
