"""Device-side counterparts of the reference's per-image pipeline functions, same names and argument
meaning, running on the HIP kernels (no CPU compute path):

    SimpleNLF / SelfNLF / CollabNLF / get_threshold   YOND_SIDD.py:13-124
    get_bias (LUT object)                              utils/isp_algos.py:98-140
    VST_Denoiser / Simple_Denoiser                     YOND_SIDD.py:238-299
    IterDenoise                                        YOND_SIDD.py:301-483 (est_type 'simple' pipelines)

Images are float32 torch tensors on a ROCm device (NumPy arrays are uploaded).  Host code here is only
control flow and O(20)-element arithmetic (threshold score, 2x2 normal equations, LUT knot grid); every
pass over pixels is a kernel launch through the C ABI of libyond_hip.so.
"""
import ctypes as C
import math

import os

import numpy as np
import torch

from . import _lib as L

NBINS = 1024

# bench.py sets this to a list to collect (tag, start event, end event) of the VST / NLE stages, recorded on the
# stream the kernels are launched on (torch's current stream)
PROF = None


class _stage:
    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        if PROF is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if PROF is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROF.append((self.tag, self.e0, e1))
        return False


def _dev(x, device=None):
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            if device is None:
                raise L.YondHipError("CPU tensor given and no device specified; the HIP path has no CPU fallback")
            x = x.to(device)
        return x.contiguous().float()
    dev = torch.device(device if device is not None else 'cuda')
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)


# ------------------------------------------------------------------------------------------------
# pack / unpack (utils/isp_ops.py:57-63)
# ------------------------------------------------------------------------------------------------
def bayer2rggb(bayer, device=None):
    b = _dev(bayer, device)
    H, W = b.shape
    out = torch.empty((H // 2, W // 2, 4), dtype=torch.float32, device=b.device)
    L.check(L.load().yond_bayer2rggb_f32(L.ptr(b), H, W, L.ptr(out), L.stream()), "yond_bayer2rggb_f32")
    return out


def rggb2bayer(rggb, device=None):
    r = _dev(rggb, device)
    h, w, c = r.shape
    if c != 4:
        raise L.YondHipError("rggb2bayer expects (h, w, 4)")
    out = torch.empty((2 * h, 2 * w), dtype=torch.float32, device=r.device)
    L.check(L.load().yond_rggb2bayer_f32(L.ptr(r), h, w, L.ptr(out), L.stream()), "yond_rggb2bayer_f32")
    return out


def rot90(x, k):
    """np.rot90(x, k, axes=(-2, -1)) of a device tensor [.., H, W] (bit exact copy kernel)."""
    x = _dev(x)
    H, W = x.shape[-2:]
    n = x.numel() // (H * W)
    k = k % 4
    out = torch.empty(x.shape[:-2] + ((W, H) if k & 1 else (H, W)), dtype=torch.float32, device=x.device)
    L.check(L.load().yond_rot90_f32(L.ptr(x), n, H, W, k, L.ptr(out), L.stream()), "yond_rot90_f32")
    return out


def get_p2d(shape, base=16):
    """utils/utils.py:246-252."""
    xb, xc, xh, xw = shape
    yh, yw = ((xh - 1) // base + 1) * base, ((xw - 1) // base + 1) * base
    dY, dX = yh - xh, yw - xw
    return (dX // 2, dX - dX // 2, dY // 2, dY - dY // 2)


# ------------------------------------------------------------------------------------------------
# scalar VST helpers (host float64; the per-pixel transforms are fused into K1 / K4)
# ------------------------------------------------------------------------------------------------
def vst_scalar(x, sigma, gain):
    """utils/isp_algos.py:5-14 for a scalar x (the reference evaluates lower = VST(0), upper = VST(scale))."""
    fz = np.float64(gain) * x + (3 / 8) * np.float64(gain) ** 2 + np.float64(sigma) ** 2
    fz = np.maximum(fz, 0)
    return 2 / np.float64(gain) * fz ** 0.5


# ------------------------------------------------------------------------------------------------
# bias LUT (utils/isp_algos.py:98-140): knot grid on the host, expectation integrals on the device
# ------------------------------------------------------------------------------------------------
def _bias_knots(ub):
    """utils/isp_algos.py:101-108 (same np.linspace calls, so the knots are bit-identical)."""
    lb = 0
    if ub < 50:
        return np.linspace(lb, ub, int((ub - lb) / 0.1) + 2)
    elif ub < 500:
        return np.concatenate((np.linspace(lb, 50, int((50 - lb) / 0.1) + 1), np.linspace(50, ub, int(ub - 50) + 2)))
    return np.concatenate((np.linspace(lb, 50, int((50 - lb) / 0.1) + 1), np.linspace(50, 500, 451),
                           np.linspace(500, ub, int(ub - 500) // 10 + 2)))


_KNOT_CACHE = {}


def _knots_on_device(ub, dev):
    """The knot grid depends only on ceil(max)+1: keep the last few grids resident (host array + float64 device
    copy) so that consecutive frames of a dataset skip the host linspace and the H2D copy."""
    key = (float(ub), str(np.asarray(ub).dtype), str(dev))       # a float32 max gives float32 knots below 50 (NumPy 2)
    hit = _KNOT_CACHE.get(key)
    if hit is None:
        lams = _bias_knots(ub)
        hit = (lams, torch.from_numpy(np.ascontiguousarray(lams, dtype=np.float64)).to(dev))
        if len(_KNOT_CACHE) >= 8:
            _KNOT_CACHE.pop(next(iter(_KNOT_CACHE)))
        _KNOT_CACHE[key] = hit
    return hit


class DeviceBiasLUT:
    """What `get_bias` returns: the interp1d knots, resident on the device.  K1 evaluates it per pixel."""

    def __init__(self, lams, x_dev, y_dev):
        self.lams = lams
        self.x = x_dev          # float64 [n]
        self.y = y_dev          # float32 [n]

    def __len__(self):
        return int(self.x.numel())

    def __call__(self, x):
        """The interp1d object called on an array (utils/isp_algos.py:128): float64 biases on the device.  (The hot path
        evaluates the LUT per pixel inside yond_pack_vst_norm_f32 instead.)"""
        xd = _dev(x, self.x.device).reshape(-1)
        out = torch.empty(xd.numel(), dtype=torch.float64, device=xd.device)
        L.check(L.load().yond_bias_eval_f32(L.ptr(xd), xd.numel(), L.ptr(self.x), L.ptr(self.y), len(self), 0, 0, 1.0, 0.0,
                                            L.ptr(out), L.stream()), "yond_bias_eval_f32")
        return out.reshape(tuple(np.shape(x)))


class BiasLUT:
    """utils/isp_algos.py:162-231: the precomputed 2-D bias table (x in e-, sigma in e-), `get_lut(x, K, sigGs)` per pixel.
    lut_path: the reference's checkpoints/bias_lut_2d.npy ((x_len, sg_len) on the grids of :168-177); or pass `table`
    with its own `x_lut` / `sg_lut` grids.  The table stays on the host (two rows are merged per (K, sigma) on the host,
    :188-194); the merged row is evaluated per pixel on the device (K1 / yond_bias_eval_f32)."""

    def __init__(self, lut_path='checkpoints/bias_lut_2d.npy', table=None, x_lut=None, sg_lut=None):
        self.bias_lut = np.load(lut_path) if table is None else np.asarray(table)
        sp = 128
        self.x_lut = np.asarray(x_lut, np.float64) if x_lut is not None else np.concatenate((
            np.linspace(0, 2 ** -4, sp, endpoint=False), np.exp(np.linspace(np.log(2 ** (-4)), np.log(2 ** 10), 14 * sp + 1))))
        self.sg_lut = np.asarray(sg_lut, np.float64) if sg_lut is not None else np.concatenate((
            np.linspace(0, 1, 200, endpoint=False), np.linspace(1, 10, 901)))
        if self.bias_lut.size != len(self.x_lut) * len(self.sg_lut):
            raise L.YondHipError(f"bias table of {self.bias_lut.size} entries does not match the {len(self.x_lut)} x {len(self.sg_lut)} grid")
        if len(self.x_lut) > 4096:
            raise L.YondHipError("the kernels hold at most 4096 LUT knots")

    def pos_interp(self, data, x):                                   # :179-186
        data = np.concatenate(([-np.inf, ], data))
        idx = np.searchsorted(data, x).clip(0, len(data) - 1)
        with np.errstate(invalid='ignore'):
            return idx - (data[idx] - x) / (data[idx] - data[idx - 1]) - 1

    def row(self, K, sigGs, device=None):
        """The 1-D LUT for (K, sigGs): (knots in DN float64, ordinates float64) on the device, or None when sigma / K lies
        outside the table (:204-212: the caller falls back to get_bias)."""
        sg = np.float64(sigGs) / np.float64(K)
        sg_pos = self.pos_interp(self.sg_lut, sg)
        sg_len = len(self.sg_lut)
        # the reference tests `sg_pos >= sg_len` (:204) and then indexes column ceil(sg_pos) (:189-194): for sigma / K just
        # above the last sigma knot (sg_len - 1 < sg_pos < sg_len) that is column sg_len -> IndexError in the reference.
        # Here such a sigma takes the get_bias fallback, like every other sigma outside the table.
        if sg_pos > sg_len - 1:
            return None
        key = (float(K), float(sigGs), str(device))
        if getattr(self, '_row_cache', None) is not None and self._row_cache[0] == key:
            return self._row_cache[1]                                  # the 32 blocks of a SIDD image share (K, sigma)
        pos = np.clip(sg_pos, 0, len(self.x_lut) - 1)                  # :189 (clips with len(x_lut) on this axis too)
        l, r = int(np.floor(pos)), int(np.ceil(pos))
        wr = pos - l
        tab = self.bias_lut.reshape(-1, sg_len)
        data = tab[:, l] * (1 - wr) + tab[:, r] * wr                  # :194
        dev = torch.device(device if device is not None else 'cuda')
        row = DeviceBiasRow(torch.from_numpy(np.ascontiguousarray(self.x_lut * np.float64(K))).to(dev),
                            torch.from_numpy(np.ascontiguousarray(data, np.float64)).to(dev), float(K), float(sigGs))
        self._row_cache = (key, row)
        return row

    def get_lut(self, x, K=1, sigGs=2, func=False, device=None):
        """:196-231: biases (float64, on the device) of an array of DN values -- or, func=True, a callable.
        sigma / K inside the table: the merged row, evaluated per pixel on the device (beyond the table's x range: the last
        ordinate, then Foi's closed form, :226-230).  Outside the table (:204-212): get_bias for images (> 1000 points) and
        for func=True, the pointwise integration get_bias_points(pho_min = 100, close_form = True) for <= 1000 points.
        func=True INSIDE the table references x_pos before assignment in the reference (:214, UnboundLocalError); here it
        returns the row as a callable over DN values, which is what that line sets out to build."""
        dev = x.device if isinstance(x, torch.Tensor) and x.is_cuda else device
        row = self.row(K, sigGs, dev)
        if row is None:                                                # sigma outside the table
            mx = x.max().item() if isinstance(x, torch.Tensor) else np.max(x)
            if func:
                return get_bias(np.float32(mx), sigGs, K, device=dev)                       # :205-206
            if int(np.prod(np.shape(x))) > 1000:
                return get_bias(np.float32(mx), sigGs, K, device=dev)(x)                    # :208-210
            return get_bias_points(x, K, sigGs, close_form=True, device=dev)                # :211-212
        return row if func else row(x)


class DeviceBiasRow:
    """One (K, sigma) row of the 2-D table on the device; K1 evaluates it per pixel (yond_pack_vst_norm_biaslut_f32)."""

    def __init__(self, x_dev, y_dev, K, sigma):
        self.x, self.y, self.K, self.sigma = x_dev, y_dev, K, sigma

    def __len__(self):
        return int(self.x.numel())

    def __call__(self, x):
        xd = _dev(x, self.x.device).reshape(-1)
        out = torch.empty(xd.numel(), dtype=torch.float64, device=xd.device)
        L.check(L.load().yond_bias_eval_f32(L.ptr(xd), xd.numel(), L.ptr(self.x), L.ptr(self.y), len(self), 1, 1, self.K, self.sigma,
                                            L.ptr(out), L.stream()), "yond_bias_eval_f32")
        return out.reshape(tuple(np.shape(x)))


def get_bias_points(lams, K, sigGs, pho_min=100, close_form=False, device=None):
    """utils/isp_algos.py:142-160 (clip=False): the bias at ARBITRARY abscissae `lams` (DN), each integrated against the
    Poisson (*) Gaussian density sampled at pho = max(int(sqrt K), pho_min) points per electron; Foi's closed form above th
    only with close_form.  Returns a device tensor of lams' shape, float64 (the reference keeps the queries' dtype: float32
    queries give float32-rounded biases there; round the result to compare bit for bit)."""
    lib = L.load()
    dev = lams.device if isinstance(lams, torch.Tensor) and lams.is_cuda else torch.device(device if device is not None else 'cuda')
    lam_h = lams.detach().cpu().numpy() if isinstance(lams, torch.Tensor) else np.asarray(lams)
    shape = lam_h.shape
    x = torch.from_numpy(np.ascontiguousarray(lam_h.reshape(-1), np.float64)).to(dev)
    n = x.numel()
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    if n == 0:
        return out.reshape(shape)
    lam_max = float(lam_h.max())
    nwg = min(256, n)
    need = int(lib.yond_bias_points_scratch(float(K), float(sigGs), int(pho_min), int(bool(close_form)), lam_max, nwg))
    if need == 0 or need * 8 > 8 << 30:
        raise L.YondHipError(f"get_bias_points(K={float(K):.4g}, sigma={float(sigGs):.4g}, pho_min={pho_min}): the integration grid is out of this build's range")
    scratch = torch.empty(need, dtype=torch.float64, device=dev)
    L.check(lib.yond_bias_points_f64(L.ptr(x), n, float(K), float(sigGs), int(pho_min), int(bool(close_form)), lam_max, None, L.ptr(out),
                                     L.ptr(scratch), need, nwg, L.stream()), "yond_bias_points_f64")
    torch.cuda.current_stream().synchronize()                # (the scratch buffer must outlive the kernel)
    if close_form:
        # The reference's index bookkeeping, reproduced as it is (:148-159): the closed-form values go to their own positions
        # (bias[lams > th] = ...), then `lams` is replaced by its integrated subset and the loop stores the i-th integrated
        # value at bias[i] -- the FIRST positions, whatever their abscissae were.  For queries on one side of th (every call
        # of the 2-D table's construction and of get_lut beyond the table) that is the identity.
        th = 50 * float(K) if float(K) < 1 else 50 * float(K) ** 0.5
        mask = torch.from_numpy(np.ascontiguousarray(lam_h.reshape(-1) > th)).to(dev)
        fixed = torch.zeros_like(out)
        fixed[mask] = out[mask]
        sub = out[~mask]
        fixed[:sub.numel()] = sub
        out = fixed
    return out.reshape(shape)


def get_bias(img=None, sigGs=25.853043, K=24.48128, device=None):
    """utils/isp_algos.py:98-140 (close_form=True, clip=False, pho_min=1).  `img`: anything with .max()
    or a scalar upper bound (the reference only uses img.max())."""
    if isinstance(img, torch.Tensor):
        device = img.device if img.is_cuda else device
        mx = np.float32(img.max().item())
    else:
        mx = np.max(img)
    ub = np.ceil(mx) + 1
    dev = torch.device(device if device is not None else 'cuda')
    lams, x_dev = _knots_on_device(ub, dev)
    y_dev = torch.empty(len(lams), dtype=torch.float32, device=dev)
    if len(lams) > 4096:
        raise L.YondHipError(f"bias LUT with {len(lams)} knots exceeds the kernel's 4096-knot LDS table")
    lib = L.load()
    rc = lib.yond_bias_lut_f64(L.ptr(x_dev), len(lams), float(K), float(sigGs), L.ptr(y_dev), L.stream())
    if rc == -2:
        # large K * sigma (14-bit frames at a digital gain): the integration tables exceed the LDS -- the same kernel with its
        # Gaussian table in a global scratch buffer, 256 workgroups striding over the knots (seconds instead of microseconds)
        nwg = 256
        need = int(lib.yond_bias_lut_big_scratch(float(K), float(sigGs), nwg))
        if need == 0 or need * 8 > 8 << 30:
            raise L.YondHipError(f"get_bias(K={float(K):.4g}, sigma={float(sigGs):.4g}): the integration grid is out of this build's range")
        scratch = torch.empty(need, dtype=torch.float64, device=dev)
        rc = lib.yond_bias_lut_big_f64(L.ptr(x_dev), len(lams), float(K), float(sigGs), L.ptr(y_dev), L.ptr(scratch), need, nwg, L.stream())
        torch.cuda.current_stream().synchronize()            # (the scratch buffer must outlive the kernel)
    L.check(rc, "yond_bias_lut_f64")
    return DeviceBiasLUT(lams, x_dev, y_dev)


# ------------------------------------------------------------------------------------------------
# noise-level estimation
# ------------------------------------------------------------------------------------------------
def _percentiles(data_flat, quants):
    lib = L.load()
    n = data_flat.numel()
    q = np.ascontiguousarray(quants, dtype=np.float64)
    ws = torch.empty(int(lib.yond_select_ws_bytes(2 * len(q))), dtype=torch.uint8, device=data_flat.device)
    out = torch.empty(len(q), dtype=torch.float64, device=data_flat.device)
    L.check(lib.yond_percentiles_f32(L.ptr(data_flat), n, C.c_void_p(q.ctypes.data), len(q), L.ptr(out), L.ptr(ws), L.stream()),
            "yond_percentiles_f32")
    return out


def _occupancy(lap, mean, ths_dev, width=None):
    """K7a over flat maps; `width` = row length of the maps (lets a lane walk down a column of the smooth maps).
    Returns the occupancy bitmap [nt][32] int32 on the device; see _occ_unpack."""
    lib = L.load()
    nt = ths_dev.numel()
    n = lap.numel()
    width = int(width) if width and n % int(width) == 0 else n
    occ = torch.empty((nt, NBINS // 32), dtype=torch.int32, device=lap.device)
    L.check(lib.yond_nlf_occupancy_f32(L.ptr(lap), L.ptr(mean), n, width, L.ptr(ths_dev), nt, L.ptr(occ), L.stream()),
            "yond_nlf_occupancy_f32")
    return occ


def _score3_device(occ, ths_dev, quants):
    """K7s: (sel = [i*, ths[i*], quants[i*], score[i*]] float64, npeaks int32) on the device."""
    lib = L.load()
    nt = ths_dev.numel()
    q = np.ascontiguousarray(quants, dtype=np.float64)
    sel = torch.empty(4, dtype=torch.float64, device=occ.device)
    npeaks = torch.empty(nt, dtype=torch.int32, device=occ.device)
    L.check(lib.yond_nlf_score3_f64(L.ptr(occ), L.ptr(ths_dev), C.c_void_p(q.ctypes.data), nt, L.ptr(sel), L.ptr(npeaks),
                                    L.stream()), "yond_nlf_score3_f64")
    return sel, npeaks


def _moments(lap, mean, var, th_dev):
    """K7b: [2][5] float64 sums {n, Sm, Sv, Smm, Smv} over lap < th (th: float64 device scalar / 1-element view)."""
    lib = L.load()
    mom = torch.empty((2, 5), dtype=torch.float64, device=lap.device)
    L.check(lib.yond_nlf_moments_f32(L.ptr(lap), L.ptr(mean), L.ptr(var), lap.numel(), L.ptr(th_dev), L.ptr(mom), L.stream()),
            "yond_nlf_moments_f32")
    return mom


def _occ_unpack(occ_words):
    """[nt][32] bitmap words (host array) -> [nt][1024] booleans."""
    w = np.ascontiguousarray(occ_words).view(np.uint32)
    return ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool).reshape(w.shape[0], -1)


def _fit_from_moments(m_all, m_ns):
    """utils/isp_algos.py:345-365 on moment sums {n, Sm, Sv, Smm, Smv}: keep the non-saturated set if it
    holds more than 1 % of the points (:349), then the 2x2 normal equations of [m, 1].[b1, b2] = v."""
    use = m_ns if m_ns[0] > 0.01 * m_all[0] else m_all
    n, sx, sy, sxx, sxy = (float(v) for v in use)
    det = n * sxx - sx * sx
    if n < 2 or det <= 1e-12 * max(n * sxx, 1e-300):
        # rank-deficient design (a constant mean map, or a single selected pixel): scipy.linalg.lstsq (:364) does not
        # raise there; return the minimum-norm solution of [m, 1].[b1, b2] = v, (b1, b2) = vbar * (mbar, 1) / (mbar^2 + 1)
        if n <= 0:
            return np.array([0.0, 0.0])
        mbar, vbar = sx / n, sy / n
        return np.array([vbar * mbar / (mbar * mbar + 1.0), vbar / (mbar * mbar + 1.0)])
    return np.array([(n * sxy - sx * sy) / det, (sxx * sy - sx * sxy) / det])


def _same(a, b):
    """Equality of two float64 results of the same formula, NaN included (a frame with NaN pixels has NaN percentiles on both
    sides: the reference carries them through; the cross-check must not turn that into an error)."""
    return a == b or (np.isnan(a) and np.isnan(b))


def _threshold_state(lap, mean, quants, ws=None):
    """K6'/K7' (nle_fast.hip): (ths float64[nq], npeaks int32[nq], sel float64[4], ws) of the two-sweep selection."""
    lib = L.load()
    width = lap.shape[-1] if lap.dim() > 1 else lap.numel()
    lap, mean = lap.reshape(-1), mean.reshape(-1)
    n = lap.numel()
    q = np.ascontiguousarray(quants, dtype=np.float64)
    qp = C.c_void_p(q.ctypes.data)
    off_ths, off_sel, off_mom, off_np, _ = _nle_layout()
    if ws is None:
        ws = _nle_workspace(n, lap.device)
        L.check(lib.yond_nle_stats_f32(L.ptr(lap), L.ptr(mean), n, int(width), qp, len(q), L.ptr(ws), L.stream()), "yond_nle_stats_f32")
    L.check(lib.yond_nle_threshold_f32(L.ptr(lap), n, qp, len(q), 1, L.ptr(ws), L.stream()), "yond_nle_threshold_f32")
    head = ws[:off_np + 4 * 32].cpu().numpy()
    nq = len(q)
    return (head[off_ths:off_ths + 8 * nq].view(np.float64).copy(), head[off_np:off_np + 4 * nq].view(np.int32).copy(),
            head[off_sel:off_sel + 32].view(np.float64).copy(), ws)


def get_threshold(data, step=5, mode='score3', print_log=False, scale=1023 - 64, _full=False):
    """YOND_SIDD.py:13-52, mode 'score3': data = (img_lap, mean) device maps (same shape).
    Returns (th, percent) like the reference; `_full` adds the internals."""
    if mode != 'score3':
        raise NotImplementedError(mode)
    lap, mean = data
    quants = np.linspace(step, 100, 100 // step, endpoint=True)
    ths, npeaks, sel, _ = _threshold_state(lap, mean, quants)
    th, pct, info = _score3(ths, quants, npeaks=npeaks)
    if info['index'] != int(sel[0]) or not _same(th, sel[1]):
        raise L.YondHipError(f"score3 mismatch: device picked {sel[0]:.0f}/{sel[1]!r}, host {info['index']}/{th!r}")
    if _full:
        return th, pct, info
    return th, pct


def _score3(ths, quants, occ=None, npeaks=None):
    """YOND_SIDD.py:37-47 from the occupancy bitmap (or from npeaks already counted on the device)."""
    if npeaks is None:
        seen = np.logical_or.accumulate(_occ_unpack(occ), axis=0)
        npeaks = seen.sum(axis=1)
    npeaks = np.asarray(npeaks).astype(np.float64)               # YOND_SIDD.py:37-43
    score = ths / (quants * npeaks)                              # :45
    i = int(np.argmin(score[1:]) + 1)                            # :46-47
    return ths[i], quants[i], dict(ths=ths, npeaks=npeaks, score=score, index=i)


_NLE_LAYOUT = None


def _nle_layout():
    global _NLE_LAYOUT
    if _NLE_LAYOUT is None:
        off = (C.c_int * 5)()
        L.check(L.load().yond_nle_state_layout(off), "yond_nle_state_layout")
        _NLE_LAYOUT = tuple(int(v) for v in off)
    return _NLE_LAYOUT


QUANTS = np.linspace(5, 100, 20, endpoint=True)            # YOND_SIDD.py:25 with step = 5


def _nle_workspace(n, dev):
    return torch.empty(int(L.load().yond_nle_ws_bytes(n)), dtype=torch.uint8, device=dev)


def _nlf_from_maps(lap, mean, var, full=False, ws=None):
    """Shared tail of SelfNLF / CollabNLF (YOND_SIDD.py:75-87 / 103-115).  Percentiles, occupancy, score3 and the
    moment sums below the selected threshold all run on the device; one host sync at the end.  `ws`: the workspace the
    producer of the maps has already filled with the level-1 statistics (fused box kernel), else a sweep does it."""
    lib = L.load()
    width = lap.shape[-1] if lap.dim() > 1 else lap.numel()
    lap, mean, var = lap.reshape(-1), mean.reshape(-1), var.reshape(-1)
    n = lap.numel()
    quants = QUANTS
    q = np.ascontiguousarray(quants, dtype=np.float64)
    qp = C.c_void_p(q.ctypes.data)
    off_ths, off_sel, off_mom, off_np, _ = _nle_layout()
    st = L.stream()
    with _stage("nle_select_score_moments"):
        if ws is None:
            ws = _nle_workspace(n, lap.device)
            L.check(lib.yond_nle_stats_f32(L.ptr(lap), L.ptr(mean), n, int(width), qp, len(q), L.ptr(ws), st), "yond_nle_stats_f32")
        L.check(lib.yond_nle_threshold_f32(L.ptr(lap), n, qp, len(q), 1, L.ptr(ws), st), "yond_nle_threshold_f32")
        L.check(lib.yond_nle_moments_f32(L.ptr(lap), L.ptr(mean), L.ptr(var), n, L.ptr(ws), st), "yond_nle_moments_f32")
    head = ws[:off_np + 4 * 32 + 4 * 64 + 8].cpu().numpy()      # the one sync (results head of the workspace)
    off_max = _nle_layout()[4]
    frame_max_key = int(head[off_max:off_max + 4].view(np.uint32)[0])
    nq = len(quants)
    ths = head[off_ths:off_ths + 8 * nq].view(np.float64).copy()
    sel_h = head[off_sel:off_sel + 32].view(np.float64).copy()
    mom_h = head[off_mom:off_mom + 80].view(np.float64).reshape(2, 5).copy()
    npeaks = head[off_np:off_np + 4 * nq].view(np.int32).copy()
    th, pct, info = _score3(ths, quants, npeaks=npeaks)
    if info['index'] != int(sel_h[0]) or not _same(th, sel_h[1]):   # host and device run the same float64 formula
        raise L.YondHipError(f"score3 mismatch: device picked {sel_h[0]:.0f}/{sel_h[1]!r}, host {info['index']}/{th!r}")
    sel = mom_h                                                  # pixels with lap < ths[i]
    if sel[0, 0] > 0:
        reg = _fit_from_moments(sel[0], sel[1])
    else:                                                        # :79-84 'no flat area'
        th_b = _percentiles(lap, [25.0])
        th_backup = float(th_b.cpu().numpy()[0])
        if th != th_backup:
            th = th_backup
            sel = _moments(lap, mean, var, th_b).cpu().numpy()
        else:                                                    # same threshold: the empty selection falls back to all
            sel = _moments(lap, mean, var, torch.full((1,), float('inf'), dtype=torch.float64, device=lap.device)).cpu().numpy()
        reg = _fit_from_moments(sel[0], sel[1])
    if full:
        info.update(th=th, percent=pct, nsel=int(sel[0, 0]))
        if frame_max_key:
            info['frame_max'] = _key2float(frame_max_key)
        return reg, info
    return reg


def _key2float(key):
    """Inverse of the kernels' order-preserving float key (nle_common.h f2key)."""
    key = int(key) & 0xFFFFFFFF
    bits = (key & 0x7FFFFFFF) if (key & 0x80000000) else (~key & 0xFFFFFFFF)
    return np.array([bits], dtype=np.uint32).view(np.float32)[0]


def SimpleNLF(lr_raw, hr_raw=None, k=29, setting=None, full=False, device=None, fused=None, box=None, _maps_only=False):
    """YOND_SIDD.py:117-124 (+ SelfNLF :62-87, CollabNLF :89-115): Bayer frame(s) -> (beta1, beta2).
    box selects the kernels that produce the three maps:
      'two-pass' (default)  the streaming box kernels with the first sweep of the threshold selection folded into the
                            producer of the lap map and the frame maximum into stage 1 (nle.hip, STATS);
      'one-pass'            one kernel per frame that keeps the B19 map in LDS (nle_fused.hip; k == 29 only);
      'plain'               the stand-alone kernels of the first version, followed by a separate statistics sweep (kept
                            for the function seam and as a cross-check).
    `fused` is the older spelling (True -> 'one-pass', False -> 'plain').  With full=True the info dict carries
    'frame_max' (float32 maximum of lr_raw) except on the 'plain' path."""
    setting = setting or {'mode': 'self'}
    lib = L.load()
    lr = _dev(lr_raw, device)
    H, W = lr.shape
    h, w = H // 2, W // 2
    tile_w = w // 32 if setting.get('SIDD_256', False) else 0
    if tile_w and w % 32:
        raise L.YondHipError("SIDD_256 needs a packed width that splits into 32 tiles")
    new = lambda: torch.empty((4, h, w), dtype=torch.float32, device=lr.device)
    mean, var, lap = new(), new(), new()
    st = L.stream()
    k2 = k // 3 * 2 + 1
    if box is None:
        box = 'two-pass' if fused is None else ('one-pass' if fused else 'plain')
    if box == 'one-pass' and k != 29:         # the one-pass kernel is built for the estimator's windows (29, 19)
        box = 'two-pass'
    if box == 'one-pass' and not L.has("yond_box_stats_self_fused_f32"):
        raise L.YondHipError("box='one-pass' needs an experiment build of the library (python -m yond_public_amd.build --experiments; "
                             "the one-pass kernels measured slower than the default producers and are not in the product)")
    if box not in ('two-pass', 'one-pass', 'plain'):
        raise ValueError(f"box={box!r}")
    ws = None
    q = np.ascontiguousarray(QUANTS, dtype=np.float64)
    qp = C.c_void_p(q.ctypes.data)
    if box != 'plain':
        ws = _nle_workspace(4 * h * w, lr.device)
    if setting['mode'] == 'self':
        with _stage("nle_box_self"):
            if box == 'one-pass':
                L.check(lib.yond_box_stats_self_fused_f32(L.ptr(lr), H, W, k, k2, tile_w, L.ptr(mean), L.ptr(var), L.ptr(lap),
                                                          qp, len(q), L.ptr(ws), st), "yond_box_stats_self_fused_f32")
            elif box == 'two-pass':
                blur2 = new()
                L.check(lib.yond_box_stats_self_stats_f32(L.ptr(lr), H, W, k, k2, tile_w, L.ptr(mean), L.ptr(var), L.ptr(blur2),
                                                          L.ptr(lap), qp, len(q), L.ptr(ws), st), "yond_box_stats_self_stats_f32")
            else:
                blur2 = new()
                L.check(lib.yond_box_stats_self1_f32(L.ptr(lr), H, W, k, k2, tile_w, L.ptr(mean), L.ptr(var), L.ptr(blur2), st),
                        "yond_box_stats_self1_f32")
                L.check(lib.yond_box_stats_self2_f32(L.ptr(blur2), h, w, k, tile_w, L.ptr(lap), st), "yond_box_stats_self2_f32")
    elif setting['mode'] == 'collab':
        hr = _dev(hr_raw, lr.device)
        if hr.shape != lr.shape:
            raise L.YondHipError("collab NLF needs noisy and denoised frames of the same shape")
        with _stage("nle_box_collab"):
            if box == 'one-pass':
                L.check(lib.yond_box_stats_collab_fused_f32(L.ptr(lr), L.ptr(hr), H, W, k, tile_w, L.ptr(mean), L.ptr(var),
                                                            L.ptr(lap), qp, len(q), L.ptr(ws), st), "yond_box_stats_collab_fused_f32")
            elif box == 'two-pass':
                L.check(lib.yond_box_stats_collab_stats_f32(L.ptr(lr), L.ptr(hr), H, W, k, tile_w, L.ptr(mean), L.ptr(var),
                                                            L.ptr(lap), qp, len(q), L.ptr(ws), st), "yond_box_stats_collab_stats_f32")
            else:
                L.check(lib.yond_box_stats_collab_f32(L.ptr(lr), L.ptr(hr), H, W, k, tile_w, L.ptr(mean), L.ptr(var), L.ptr(lap), st),
                        "yond_box_stats_collab_f32")
    else:
        raise NotImplementedError(setting['mode'])
    if _maps_only:
        return lap, mean, var, ws
    return _nlf_from_maps(lap, mean, var, full, ws=ws)


# ------------------------------------------------------------------------------------------------
# VST -> denoiser -> inverse VST  (YOND_SIDD.py:238-299)
# ------------------------------------------------------------------------------------------------
def _plan_of(net, device):
    mod = net.module if hasattr(net, 'module') else net           # nn.DataParallel wrapper
    if not hasattr(mod, '_get_plan'):
        raise L.YondHipError(f"{type(mod).__name__} is not a yond_public_amd.archs denoiser")
    return mod._get_plan(device)


class _Guard:
    """Range guard of the half-precision operand paths (engine.DenoiserPlan.uses_half_operands): the convolution kernels
    report a staged activation outside fp16's range in a device word; `finish` waits for the forward it watched, and if
    the word is set re-runs that forward on the fp32-input MFMA kernels -- an overflow never reaches the caller as a
    silent inf."""

    def __init__(self, plan, slot):
        self.plan, self.slot = plan, slot
        if not hasattr(plan, '_flag_host'):
            plan._flag_host = torch.zeros(12, dtype=torch.int32).pin_memory()
        plan.begin_guard(slot)

    def arm(self, rerun):
        """Queue the read-back of the status word behind the forward; `rerun()` recomputes the result strictly."""
        self.rerun = rerun
        self.plan._flag_host[self.slot:self.slot + 1].copy_(self.plan.status[self.slot:self.slot + 1], non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        return self

    def tripped(self):
        """True if the watched forward staged an activation outside fp16's range (waits for it)."""
        self.event.synchronize()
        return bool(int(self.plan._flag_host[self.slot]) & 1)

    def finish(self):
        """None if the forward stayed in range, else the strictly recomputed result."""
        if not self.tripped():
            return None
        import warnings
        warnings.warn("an activation left fp16's range (|a| > 65504) in the split-operand convolution path: "
                      "this forward is recomputed on the fp32-input MFMA kernels")
        self.plan.strict = True
        try:
            return self.rerun()
        finally:
            self.plan.strict = False


def VST_Denoiser(lr_raw, p, net, arch, bias_corr='pre', bias_func=None, vst_type='exact', clip01=False, device=None,
                 lr_max=None, guard=True, guard_slot=0, biaslut=None):
    """YOND_SIDD.py:250-299 for the network denoisers.  lr_raw: Bayer [H][W] -- or a stack [B][H][W] of equally
    sized frames that go through ONE batched forward instead of B batch-1 calls: the 32 blocks of a SIDD image, which
    share (gain, sigma) and the bias LUT (:392-407), or B independent frames (BASELINE cfg 4), for which `p`,
    `bias_func` and `lr_max` are lists with one entry per frame.  p: dict with scale, gain, sigma; returns the
    denoised frame(s) as a device tensor.  `clip01` folds the caller's .clip(0,1); `lr_max` (float32 max of the
    frame) spares the device reduction + sync when the caller already has it.  guard: True -- wait for the forward and
    recompute it on the fp32-input MFMA kernels if an activation left fp16's range (see _Guard); 'defer' -- return
    (result, guard) and let the caller call guard.finish() later; False -- no check."""
    lib = L.load()
    lr = _dev(lr_raw, device)
    single = lr.dim() == 2
    if single:
        lr = lr[None]
    B, H, W = lr.shape
    h, w = H // 2, W // 2
    per_frame = isinstance(p, (list, tuple))
    ps = list(p) if per_frame else [p] * B
    if len(ps) != B:
        raise L.YondHipError(f"{len(ps)} parameter sets for {B} frames")
    if bias_corr not in (None, 'pre'):
        raise NotImplementedError(f"bias_corr={bias_corr!r} (the reference's 'post' branch is commented out)")
    funcs = list(bias_func) if isinstance(bias_func, (list, tuple)) else [bias_func] * B
    maxes = list(lr_max) if isinstance(lr_max, (list, tuple)) else [lr_max] * B
    if bias_corr is not None and biaslut is not None:                 # :258-259: the 2-D table instead of get_bias
        for i in range(B):
            if funcs[i] is None:
                funcs[i] = biaslut.row(ps[i]['gain'], ps[i]['sigma'], lr.device)    # None: sigma outside -> get_bias below
    if bias_corr is not None and any(f is None for f in funcs):
        if not (single or per_frame or biaslut is not None):
            raise L.YondHipError("a stack of frames needs the shared bias LUT (the reference builds one per image, :392-397)")
        for i in range(B):
            if funcs[i] is None:
                mxv = maxes[i] if maxes[i] is not None else _frame_max(lr[i]).item()
                mx = np.float32(mxv) * np.float32(ps[i]['scale'])    # lr_rggb.max() of the float32 product (:256)
                funcs[i] = get_bias(mx, np.float64(ps[i]['sigma']), np.float64(ps[i]['gain']), device=lr.device)
    p2d = get_p2d((1, 4, h, w), base=32)
    Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
    x4 = torch.empty((B, Hp, Wp, 4), dtype=torch.float32, device=lr.device)
    img_max = torch.empty(B, dtype=torch.float32, device=lr.device)
    st = L.stream()
    consts, t_host = [], []
    # a stack whose frames share p and the LUT (the 32 blocks of a SIDD image): K1 and K4 as ONE launch each
    shared = B > 1 and not per_frame and all(f is funcs[0] for f in funcs)
    if shared and not lr.is_contiguous():
        lr = lr.contiguous()
    with _stage("vst_pack"):
        if shared:
            scale, gain, sigma = float(ps[0]['scale']), np.float64(ps[0]['gain']), np.float64(ps[0]['sigma'])
            lower, upper = vst_scalar(0, sigma, gain), vst_scalar(scale, sigma, gain)
            consts = [(scale, gain, sigma, lower, upper)] * B
            t_host = [float(np.float32(1 / (upper - lower) * (1.03 if bias_corr == 'pre' else 1.00)))] * B     # :284-285
            f = funcs[0]
            lut_n = len(f) if bias_corr is not None else 0
            L.check(lib.yond_pack_vst_norm_batch_f32(L.ptr(lr), B, H, W, L.ptr(x4), p2d[0], p2d[1], p2d[2], p2d[3], scale, float(gain),
                                                     float(sigma), float(lower), float(upper), L.ptr(f.x) if lut_n else None,
                                                     L.ptr(f.y) if lut_n else None, lut_n, int(bool(lut_n) and isinstance(f, DeviceBiasRow)),
                                                     L.ptr(img_max), st), "yond_pack_vst_norm_batch_f32")
        for i in range(0 if shared else B):
            scale, gain, sigma = float(ps[i]['scale']), np.float64(ps[i]['gain']), np.float64(ps[i]['sigma'])
            lower, upper = vst_scalar(0, sigma, gain), vst_scalar(scale, sigma, gain)
            consts.append((scale, gain, sigma, lower, upper))
            t_host.append(float(np.float32(1 / (upper - lower) * (1.03 if bias_corr == 'pre' else 1.00))))   # :284-285
            f = funcs[i]
            lut_n = len(f) if bias_corr is not None else 0
            if lut_n and isinstance(f, DeviceBiasRow):
                L.check(lib.yond_pack_vst_norm_biaslut_f32(L.ptr(lr[i]), H, W, L.ptr(x4[i]), p2d[0], p2d[1], p2d[2], p2d[3], scale,
                                                           float(gain), float(sigma), float(lower), float(upper), L.ptr(f.x),
                                                           L.ptr(f.y), lut_n, L.ptr(img_max[i:i + 1]), st), "yond_pack_vst_norm_biaslut_f32")
                continue
            L.check(lib.yond_pack_vst_norm_f32(L.ptr(lr[i]), H, W, L.ptr(x4[i]), p2d[0], p2d[1], p2d[2], p2d[3], 1, scale,
                                               float(gain), float(sigma), float(lower), float(upper),
                                               L.ptr(f.x) if lut_n else None, L.ptr(f.y) if lut_n else None, lut_n,
                                               L.ptr(img_max[i:i + 1]), st), "yond_pack_vst_norm_f32")
    plan = _plan_of(net, lr.device)
    t_dev = None
    if 'guided' in arch:
        t_dev = torch.tensor(t_host, dtype=torch.float32).to(lr.device, non_blocking=True) if per_frame else \
            torch.full((B,), t_host[0], dtype=torch.float32, device=lr.device)
    exact_inverse = bias_corr is None and vst_type == 'exact'

    def forward_and_invert():
        y4 = plan.forward_nhwc4(x4, t_dev, ub=img_max)
        out = torch.empty((B, H, W), dtype=torch.float32, device=lr.device)
        with _stage("ivst_unpack"):
            if shared:
                scale, gain, sigma, lower, upper = consts[0]
                L.check(lib.yond_denorm_ivst_unpack_batch_f32(L.ptr(y4), B, Hp, Wp, p2d[2], p2d[0], h, w, L.ptr(out),
                                                              2 if exact_inverse else 1, scale, float(gain), float(sigma), float(lower),
                                                              float(upper), int(clip01), st), "yond_denorm_ivst_unpack_batch_f32")
            for i in range(0 if shared else B):
                scale, gain, sigma, lower, upper = consts[i]
                L.check(lib.yond_denorm_ivst_unpack_f32(L.ptr(y4[i]), Hp, Wp, p2d[2], p2d[0], h, w, L.ptr(out[i]),
                                                        2 if exact_inverse else 1, scale, float(gain), float(sigma), float(lower),
                                                        float(upper), int(clip01), st), "yond_denorm_ivst_unpack_f32")
        return out[0] if single else out

    watch = _Guard(plan, guard_slot) if (guard and plan.uses_half_operands()) else None
    out = forward_and_invert()
    if watch is None:
        return (out, None) if guard == 'defer' else out
    watch.arm(forward_and_invert)
    if guard == 'defer':
        return out, watch
    redo = watch.finish()
    return out if redo is None else redo


# ------------------------------------------------------------------------------------------------
# The per-frame parameter chain on the device (csrc/frame_chain.hip): estimator -> (K, sigma, lower, upper, t, knots) -> bias LUT
# -> K1 -> network -> K4 without a host round trip.  Used by IterDenoise for bare full frames with the 1-D bias LUT (the
# BASELINE cfg-2 / cfg-5 configuration); every other configuration keeps the host-side chain above.
# ------------------------------------------------------------------------------------------------
LUT_CAP = 1536                      # knots the chain's buffers hold (a [0, 1] frame at scale 959 needs 1000)
PRM = dict(beta1=0, beta2=1, gain=2, sigma=3, lo=4, hi=5, nsr=6, t=7, flags=8, lut_n=9, nsel=10, th=11, pct=12, frame_max=13)
PRM_NO_FLAT_AREA, PRM_LUT_CAPACITY, PRM_ROUND_ABORTED, PRM_BAD_ESTIMATE = 1, 2, 4, 8


class _ChainBuffers:
    """Device buffers of one round of the chain (kept per device and round: nothing is allocated per frame)."""

    def __init__(self, dev):
        lib = L.load()
        self.prm = torch.zeros(16, dtype=torch.float64, device=dev)
        self.t = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lut_x = torch.zeros(LUT_CAP, dtype=torch.float64, device=dev)
        self.lut_y = torch.zeros(LUT_CAP, dtype=torch.float32, device=dev)
        self.lut_ws = torch.zeros(int(lib.yond_lut_ws_bytes(LUT_CAP)), dtype=torch.uint8, device=dev)
        self.img_max = torch.zeros(1, dtype=torch.float32, device=dev)
        self.prm_host = torch.zeros(16, dtype=torch.float64).pin_memory()
        torch.cuda.synchronize(dev)        # (the zero fills above ran on the creating stream; the buffers are written from others)


_CHAIN_BUFFERS = {}


def _chain_buffers(dev, slot):
    key = (str(dev), slot)
    if key not in _CHAIN_BUFFERS:
        _CHAIN_BUFFERS[key] = _ChainBuffers(dev)
    return _CHAIN_BUFFERS[key]


def chain_applies(lr, net, arch, pipe, biaslut=None):
    """The configurations the device chain covers: one bare Bayer frame, full_dn, bias_corr 'pre' with the 1-D LUT."""
    return (isinstance(lr, torch.Tensor) and lr.is_cuda and lr.dim() == 2 and bool(pipe.get('full_dn', False)) and biaslut is None
            and pipe.get('bias_corr', 'pre') == 'pre' and pipe.get('full_est', True) and DEVICE_CHAIN
            and 'simple' in str(pipe.get('est_type', 'simple')) and 'cal_est' not in pipe)


DEVICE_CHAIN = True                 # (module attribute: tools / tests switch the host-side chain back on for A/B)
CHAIN_ONE_LAUNCH = False            # parameters + bias LUT + table as ONE launch (yond_frame_chain_f64): same results, fewer launches, but
                                    # measured 36 us against 31 us for the three launches (every workgroup re-derives the parameters,
                                    # the table is a serial tail) -- off by default
CHAIN_K1 = True                     # the chain's own K1 (yond_pack_vst_norm_chain_f32: 29.5 us against 45.5); False: the general kernel


def _chain_estimate(lr, hr, mode, pipe, p, buf, lr_max_dev=None):
    """First half of a round of the device chain, queued on the current stream: estimator (self / collab) -> threshold ->
    moments -> yond_frame_params_f64 -> bias LUT -> prepared table, all into `buf`."""
    lib = L.load()
    k = pipe.get('k', 29)
    W = lr.shape[1]
    scale_est, scale = float(p['wp'] - p['bl']), float(p['scale'])       # :356 / :251 (equal for ratio 1)
    st = L.stream()
    setting = {'mode': mode}
    if mode == 'collab':
        setting['SIDD_256'] = bool(pipe.get('collab_sidd256', (W // 2) % 32 == 0))
    lap, mean, var, ws = SimpleNLF(lr, hr, k=k, setting=setting, _maps_only=True)
    n = lap.numel()
    q = np.ascontiguousarray(QUANTS, dtype=np.float64)
    qp = C.c_void_p(q.ctypes.data)
    with _stage("nle_select_score_moments"):
        L.check(lib.yond_nle_threshold_f32(L.ptr(lap.reshape(-1)), n, qp, len(q), 1, L.ptr(ws), st), "yond_nle_threshold_f32")
        L.check(lib.yond_nle_moments_f32(L.ptr(lap.reshape(-1)), L.ptr(mean.reshape(-1)), L.ptr(var.reshape(-1)), n, L.ptr(ws), st),
                "yond_nle_moments_f32")
    with _stage("frame_params_lut"):
        # (round 2 reads the frame maximum round 1's estimator collected: the collab kernels read the same noisy frame)
        mx = L.ptr(lr_max_dev) if lr_max_dev is not None else None
        md = 0 if mode == 'self' else 1
        if CHAIN_ONE_LAUNCH:
            L.check(lib.yond_frame_chain_f64(L.ptr(ws), mx, md, scale_est, scale, 1.03, LUT_CAP, L.ptr(buf.prm), L.ptr(buf.t), L.ptr(buf.lut_x),
                                             L.ptr(buf.lut_y), L.ptr(buf.lut_ws), st), "yond_frame_chain_f64")
        else:
            L.check(lib.yond_frame_params_f64(L.ptr(ws), mx, md, scale_est, scale, 1.03, LUT_CAP, L.ptr(buf.prm), L.ptr(buf.t), L.ptr(buf.lut_x), st),
                    "yond_frame_params_f64")
            L.check(lib.yond_bias_lut_dev_f64(L.ptr(buf.lut_x), LUT_CAP, L.ptr(buf.prm), L.ptr(buf.lut_y), st), "yond_bias_lut_dev_f64")
            L.check(lib.yond_lut_table_f64(L.ptr(buf.lut_x), L.ptr(buf.lut_y), -1, L.ptr(buf.prm), L.ptr(buf.lut_ws), st), "yond_lut_table_f64")
    buf.ws = ws                                              # (kept alive until the buffers' next use)


def _chain_denoise(lr, net, arch, p, buf, guard_slot):
    """Second half: K1 -> network -> K4 with the constants of `buf`, queued on the current stream.  Returns (frame, guard)."""
    lib = L.load()
    H, W = lr.shape
    h, w = H // 2, W // 2
    scale = float(p['scale'])
    st = L.stream()
    p2d = get_p2d((1, 4, h, w), base=32)
    Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
    x4 = torch.empty((1, Hp, Wp, 4), dtype=torch.float32, device=lr.device)
    with _stage("vst_pack"):
        k1 = lib.yond_pack_vst_norm_chain_f32 if CHAIN_K1 else lib.yond_pack_vst_norm_dev_f32
        L.check(k1(L.ptr(lr), H, W, L.ptr(x4), p2d[0], p2d[1], p2d[2], p2d[3], scale, L.ptr(buf.prm), L.ptr(buf.lut_ws), LUT_CAP,
                   L.ptr(buf.img_max), st), "yond_pack_vst_norm_chain_f32")
    plan = _plan_of(net, lr.device)
    t_dev = buf.t if 'guided' in arch else None

    def forward_and_invert():
        y4 = plan.forward_nhwc4(x4, t_dev, ub=buf.img_max)
        out = torch.empty((H, W), dtype=torch.float32, device=lr.device)
        with _stage("ivst_unpack"):
            L.check(lib.yond_denorm_ivst_unpack_dev_f32(L.ptr(y4), Hp, Wp, p2d[2], p2d[0], h, w, L.ptr(out), 1, scale, L.ptr(buf.prm),
                                                        1, st), "yond_denorm_ivst_unpack_dev_f32")
        return out

    watch = _Guard(plan, guard_slot) if plan.uses_half_operands() else None
    out = forward_and_invert()
    if watch is not None:
        watch.arm(forward_and_invert)
    buf.prm_host.copy_(buf.prm, non_blocking=True)       # read by the caller behind its synchronisation
    return out, watch


def _chain_denoise_frames(frames, net, arch, p, bufs, guard_slot):
    """_chain_denoise for B equally sized full frames, each with ITS parameter block and table (bufs[b]): K1 per frame into its slice of ONE network
    input -> one batch-B forward (every item with its own maximum and t) -> K4 per frame.  Returns ([B] of [H][W], guard)."""
    lib = L.load()
    B = len(frames)
    H, W = frames[0].shape
    dev = frames[0].device
    h, w = H // 2, W // 2
    scale = float(p['scale'])
    st = L.stream()
    p2d = get_p2d((1, 4, h, w), base=32)
    Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
    x4 = torch.empty((B, Hp, Wp, 4), dtype=torch.float32, device=dev)
    img_max = torch.empty(B, dtype=torch.float32, device=dev)
    k1 = lib.yond_pack_vst_norm_chain_f32 if CHAIN_K1 else lib.yond_pack_vst_norm_dev_f32
    with _stage("vst_pack"):
        for b, (lr, buf) in enumerate(zip(frames, bufs)):
            L.check(k1(L.ptr(lr), H, W, L.ptr(x4[b]), p2d[0], p2d[1], p2d[2], p2d[3], scale, L.ptr(buf.prm), L.ptr(buf.lut_ws), LUT_CAP,
                       L.ptr(img_max[b:b + 1]), st), "yond_pack_vst_norm_chain_f32")
    plan = _plan_of(net, dev)
    t_dev = torch.cat([buf.t for buf in bufs]).contiguous() if 'guided' in arch else None

    def forward_and_invert():
        y4 = plan.forward_nhwc4(x4, t_dev, ub=img_max)
        outs = []
        with _stage("ivst_unpack"):
            for b, buf in enumerate(bufs):
                out = torch.empty((H, W), dtype=torch.float32, device=dev)
                L.check(lib.yond_denorm_ivst_unpack_dev_f32(L.ptr(y4[b]), Hp, Wp, p2d[2], p2d[0], h, w, L.ptr(out), 1, scale, L.ptr(buf.prm), 1, st),
                        "yond_denorm_ivst_unpack_dev_f32")
                outs.append(out)
        return outs

    watch = _Guard(plan, guard_slot) if plan.uses_half_operands() else None
    outs = forward_and_invert()
    if watch is not None:
        watch.arm(forward_and_invert)
    for buf in bufs:
        buf.prm_host.copy_(buf.prm, non_blocking=True)
    return outs, watch


def _chain_round(lr, hr, mode, net, arch, pipe, p, slot, lr_max_dev=None, vst_type='exact'):
    """One round of IterDenoise for a bare frame, queued without any host synchronisation:
    estimator (self / collab) -> yond_frame_params_f64 -> bias LUT -> table -> K1 -> network -> K4.
    Returns (denoised frame [H][W] on the device, buffers whose .prm block describes the round, range guard or None)."""
    buf = _chain_buffers(lr.device, slot)
    _chain_estimate(lr, hr, mode, pipe, p, buf, lr_max_dev)
    out, watch = _chain_denoise(lr, net, arch, p, buf, slot)
    return out, buf, watch


def _chain_result(buf):
    """The round's parameter block as host numbers (the caller has synchronised): (reg, (K, sigma), flags, info)."""
    v = buf.prm_host.numpy()
    flags = int(v[PRM['flags']])
    reg = (np.float64(v[PRM['beta1']]), np.float64(v[PRM['beta2']]))
    info = dict(th=float(v[PRM['th']]), percent=float(v[PRM['pct']]), nsel=int(v[PRM['nsel']]), frame_max=np.float32(v[PRM['frame_max']]))
    return reg, (np.float64(v[PRM['gain']]), np.float64(v[PRM['sigma']])), flags, info


def _iter_denoise_chain(lr, net, arch, pipe, p, log=None):
    """IterDenoise for a bare full frame on the device chain: both rounds are queued back to back (round 2 speculatively: its
    estimate needs round 1's output, not its numbers) and ONE synchronisation at the end delivers the parameter blocks.
    Returns None when a round took a branch the chain leaves to the host (flags), and the caller runs the host-side path."""
    out1, b1, g1 = _chain_round(lr, None, 'self', net, arch, pipe, p, slot=0)
    two = pipe.get('iter', 'iter') == 'iter' and pipe.get('max_iter', 1) >= 1
    if two and pipe.get('max_iter', 1) > 1:
        return None
    if two:
        mx = torch.empty(1, dtype=torch.float32, device=lr.device)
        mx.copy_(b1.prm[PRM['frame_max']:PRM['frame_max'] + 1])           # float64 holding the float32 maximum -> float32
        out2, b2, g2 = _chain_round(lr, out1, 'collab', net, arch, pipe, p, slot=1, lr_max_dev=mx)
    torch.cuda.current_stream().synchronize()
    reg1, par1, fl1, info1 = _chain_result(b1)
    if fl1 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY | PRM_BAD_ESTIMATE):
        return None
    if g1 is not None and g1.tripped():
        return None                                      # an activation left fp16's range: the guarded host path recomputes
    if log:
        log(f"Self Est: K={par1[0]:.4f}, b={par1[1]:.4f} (beta1={reg1[0]:.3e}, beta2={reg1[1]:.3e})")
    raw_dns, regs, params = [out1], [reg1], [par1]
    if two:
        reg2, par2, fl2, info2 = _chain_result(b2)
        if fl2 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY):
            return None
        if log:
            log(f"Iter 1 Est: K={par2[0]:.4f}, sigma={par2[1]:.4f} (beta1={reg2[0]:.3e}, beta2={reg2[1]:.3e})")
        if not (fl2 & PRM_ROUND_ABORTED):                # :445-447: beta1 < 0 ends the image after round 1
            if fl2 & PRM_BAD_ESTIMATE:
                return None
            if g2 is not None and g2.tripped():
                return None
            raw_dns.append(out2)
            regs.append(reg2)
            params.append(par2)
    return dict(raw_dns=raw_dns, regs=regs, params=params, nle_info=info1)


def chain_applies_sidd(lr, lr_full, net, arch, pipe, p, biaslut=None):
    """The SIDD layout on the device chain: the [32][256][256] stack denoised block-wise (one batch-32 forward per round), the estimate
    on the stack's concatenation or on the full frame lr_full, bias_corr 'pre' with the 1-D LUT, est_type 'simple', no rot_cfa."""
    return (isinstance(lr, torch.Tensor) and lr.is_cuda and lr.dim() == 3 and lr.shape[0] == 32 and not pipe.get('full_dn', False)
            and biaslut is None and pipe.get('bias_corr', 'pre') == 'pre' and pipe.get('full_est', True) and DEVICE_CHAIN and CHAIN_SIDD
            and 'simple' in str(pipe.get('est_type', 'simple')) and 'cal_est' not in pipe and 'rot_cfa' not in p
            and (lr_full is None or (isinstance(lr_full, torch.Tensor) and lr_full.is_cuda and lr_full.dim() == 2)))


CHAIN_SIDD = True                   # (False: the SIDD layout on the host-side chain, for A/B)


def _chain_denoise_blocks(blocks, net, arch, p, buf, guard_slot):
    """_chain_denoise for B blocks that share the round's parameter block and table: batched K1 -> ONE batch-B forward -> batched K4
    (YOND_SIDD.py:392-407 runs the blocks one by one with the same p and bias_func).  Returns ([B][H][W], guard)."""
    outs, watch = _chain_denoise_blocks_group([blocks], net, arch, p, [buf], guard_slot)
    return outs[0], watch


def _chain_denoise_blocks_group(blocks_list, net, arch, p, bufs, guard_slot):
    """The same for G images at once: image g's B blocks are stabilised with ITS parameter block and table (bufs[g]: one batched K1 per image
    into its slice of the network input), the G x B blocks go through ONE forward (images are independent, YOND_SIDD.py:507-514; every block
    carries its own maximum and its image's t -- the deep levels of a batch-32 forward of 128 x 128 blocks are 16 x 16 and 8 x 8 pixels and
    leave most of the chip idle), one batched K4 per image.  Per block the kernels and their arguments are those of the one-image call
    (the forward of a block is bit-identical whatever batch it rides in: tests/test_hip_eval.py).  Returns ([G] of [B][H][W], guard of the forward)."""
    lib = L.load()
    G = len(blocks_list)
    B, H, W = blocks_list[0].shape
    dev = blocks_list[0].device
    h, w = H // 2, W // 2
    scale = float(p['scale'])
    st = L.stream()
    p2d = get_p2d((1, 4, h, w), base=32)
    Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
    x4 = torch.empty((G * B, Hp, Wp, 4), dtype=torch.float32, device=dev)
    img_max = torch.empty(G * B, dtype=torch.float32, device=dev)
    with _stage("vst_pack"):
        for g, (blocks, buf) in enumerate(zip(blocks_list, bufs)):
            L.check(lib.yond_pack_vst_norm_batch_dev_f32(L.ptr(blocks), B, H, W, L.ptr(x4[g * B:(g + 1) * B]), p2d[0], p2d[1], p2d[2], p2d[3], scale,
                                                         L.ptr(buf.prm), L.ptr(buf.lut_ws), LUT_CAP, L.ptr(img_max[g * B:(g + 1) * B]), st),
                    "yond_pack_vst_norm_batch_dev_f32")
    plan = _plan_of(net, dev)
    t_dev = None
    if 'guided' in arch:
        t_dev = (bufs[0].t.expand(B) if G == 1 else torch.cat([buf.t.expand(B) for buf in bufs])).contiguous()

    def forward_and_invert():
        y4 = plan.forward_nhwc4(x4, t_dev, ub=img_max)
        outs = []
        with _stage("ivst_unpack"):
            for g, buf in enumerate(bufs):
                out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
                L.check(lib.yond_denorm_ivst_unpack_batch_dev_f32(L.ptr(y4[g * B:(g + 1) * B]), B, Hp, Wp, p2d[2], p2d[0], h, w, L.ptr(out), 1, scale,
                                                                  L.ptr(buf.prm), 1, st), "yond_denorm_ivst_unpack_batch_dev_f32")
                outs.append(out)
        return outs

    watch = _Guard(plan, guard_slot) if plan.uses_half_operands() else None
    outs = forward_and_invert()
    if watch is not None:
        watch.arm(forward_and_invert)
    for buf in bufs:
        buf.prm_host.copy_(buf.prm, non_blocking=True)
    return outs, watch


def _iter_denoise_chain_sidd(blocks, lr_full, net, arch, pipe, p, log=None):
    """IterDenoise for the SIDD layout on the device chain (YOND_SIDD.py:312-470, est_type 'simple', block-wise denoising): round 1's
    estimate on lr_full (or the blocks' concatenation, :340), the LUT grid from the blocks' maximum (:393), the 32 blocks through one
    batch-32 forward; round 2's collaborative estimate on the concatenations with the SIDD_256 re-tiling (:431); both rounds queued
    back to back, ONE synchronisation.  None when a round took a branch the chain leaves to the host-side path."""
    return _iter_denoise_chain_sidd_group([(blocks, lr_full)], net, arch, pipe, p, log=log)[0]


def _iter_denoise_chain_sidd_group(items, net, arch, pipe, p, log=None):
    """_iter_denoise_chain_sidd for G images [(blocks, lr_full)] as ONE group: every image keeps its own estimates, parameter blocks and
    tables (round 1: self estimate on its full frame; round 2: collaborative estimate on its own concatenations); round 1 of all G images is
    ONE batch-(32 G) forward, round 2 likewise (speculatively for every image: whether an image's guard ends it after round 1, :445-447, is
    read from its parameter block afterwards, as in the one-image chain); ONE synchronisation for the group.  Returns a list of G results
    (None for an image that took a branch the chain leaves to the host-side path).  An image's result is that of its one-image call (the
    group only changes which images share a launch; what differs is what two runs of ONE image differ by: the estimator's float64 moment
    sums are accumulated with atomics -- tests/test_hip_eval.py)."""
    two = pipe.get('iter', 'iter') == 'iter' and pipe.get('max_iter', 1) >= 1
    if two and pipe.get('max_iter', 1) > 1:
        return [None] * len(items)
    dev = items[0][0].device
    G = len(items)
    key = (lambda g, r: r) if G == 1 else (lambda g, r: ('group', g, r))       # (one image: the chain's two resident buffer sets)
    st = []
    for g, (blocks, lr_full) in enumerate(items):
        blocks = blocks.contiguous()
        lr_cat = torch.cat(list(blocks), dim=-1).contiguous()                              # :315
        mx = _frame_max(lr_cat)                                                            # upper_bound = lr_raw.max() * (wp - bl), :393
        b1 = _chain_buffers(dev, key(g, 0))
        _chain_estimate(lr_cat if lr_full is None else lr_full, None, 'self', pipe, p, b1, lr_max_dev=mx)
        st.append(dict(blocks=blocks, lr_cat=lr_cat, mx=mx, b1=b1))
    o1, g1 = _chain_denoise_blocks_group([q['blocks'] for q in st], net, arch, p, [q['b1'] for q in st], 0)
    for q, o in zip(st, o1):
        q['out1'] = torch.cat(list(o), dim=-1).contiguous()                                # :408
    if two:
        pipe2 = dict(pipe, collab_sidd256=pipe.get('collab_sidd256', True))
        for g, q in enumerate(st):
            q['b2'] = _chain_buffers(dev, key(g, 1))
            _chain_estimate(q['lr_cat'], q['out1'], 'collab', pipe2, p, q['b2'], lr_max_dev=q['mx'])
        o2, g2 = _chain_denoise_blocks_group([q['blocks'] for q in st], net, arch, p, [q['b2'] for q in st], 1)
        for q, o in zip(st, o2):
            q['out2'] = torch.cat(list(o), dim=-1).contiguous()
    torch.cuda.current_stream().synchronize()
    trip1 = g1 is not None and g1.tripped()
    trip2 = two and g2 is not None and g2.tripped()
    results = []
    for g, q in enumerate(st):
        tag = f"[image {g} of the group] " if G > 1 else ""
        reg1, par1, fl1, info1 = _chain_result(q['b1'])
        if fl1 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY | PRM_BAD_ESTIMATE) or trip1:
            results.append(None)
            continue
        if log:
            log(f"{tag}Self Est: K={par1[0]:.4f}, b={par1[1]:.4f} (beta1={reg1[0]:.3e}, beta2={reg1[1]:.3e})")
        raw_dns, regs, params = [q['out1']], [reg1], [par1]
        if two:
            reg2, par2, fl2, info2 = _chain_result(q['b2'])
            if fl2 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY):
                results.append(None)
                continue
            if log:
                log(f"{tag}Iter 1 Est: K={par2[0]:.4f}, sigma={par2[1]:.4f} (beta1={reg2[0]:.3e}, beta2={reg2[1]:.3e})")
            if not (fl2 & PRM_ROUND_ABORTED):                # :445-447: beta1 < 0 ends the image after round 1
                if fl2 & PRM_BAD_ESTIMATE or trip2:
                    results.append(None)
                    continue
                raw_dns.append(q['out2'])
                regs.append(reg2)
                params.append(par2)
        results.append(dict(raw_dns=raw_dns, regs=regs, params=params, nle_info=info1))
    return results


def _frame_max(x):
    """Maximum of a device tensor as a 1-element device tensor (yond_image_max_f32: two launches, deterministic)."""
    lib = L.load()
    x = x.contiguous()
    partial = torch.empty(256, dtype=torch.float32, device=x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    L.check(lib.yond_image_max_f32(L.ptr(x), 1, x.numel(), L.ptr(partial), L.ptr(out), L.stream()), "yond_image_max_f32")
    return out


def Simple_Denoiser(lr_raw, net, device=None):
    """YOND_SIDD.py:238-248: pack, reflect-pad, clamp, net(x), clamp, crop, unpack (no VST)."""
    lib = L.load()
    lr = _dev(lr_raw, device)
    H, W = lr.shape
    h, w = H // 2, W // 2
    p2d = get_p2d((1, 4, h, w), base=32)
    Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
    x4 = torch.empty((1, Hp, Wp, 4), dtype=torch.float32, device=lr.device)
    img_max = torch.empty(1, dtype=torch.float32, device=lr.device)
    st = L.stream()
    L.check(lib.yond_pack_vst_norm_f32(L.ptr(lr), H, W, L.ptr(x4), p2d[0], p2d[1], p2d[2], p2d[3], 0, 1.0, 1.0, 0.0, 0.0, 1.0,
                                       None, None, 0, L.ptr(img_max), st), "yond_pack_vst_norm_f32")
    plan = _plan_of(net, lr.device)
    if plan.guided:
        raise L.YondHipError("Simple_Denoiser calls net(x) without a noise level (YOND_SIDD.py:244)")
    def forward_and_unpack():
        y4 = plan.forward_nhwc4(x4, None, ub=img_max)
        out = torch.empty((H, W), dtype=torch.float32, device=lr.device)
        L.check(lib.yond_denorm_ivst_unpack_f32(L.ptr(y4), Hp, Wp, p2d[2], p2d[0], h, w, L.ptr(out), 0, 1.0, 1.0, 0.0, 0.0, 1.0, 0, st),
                "yond_denorm_ivst_unpack_f32")
        return out

    watch = _Guard(plan, 0) if plan.uses_half_operands() else None
    out = forward_and_unpack()
    if watch is not None:
        redo = watch.arm(forward_and_unpack).finish()
        out = out if redo is None else redo
    return out


# ------------------------------------------------------------------------------------------------
# IterDenoise (YOND_SIDD.py:301-483)
# ------------------------------------------------------------------------------------------------
def default_params():
    """YOND_SIDD.py:503-505."""
    p = {'wp': 1023, 'bl': 64, 'ratio': 1, 'gain': 1, 'sigma': 0}
    p['scale'] = (p['wp'] - p['bl']) / p['ratio']
    return p


def file_estimate(pipe, est=None):
    """The est_types that READ an estimate instead of computing one (YOND_SIDD.py:316-337): (beta1, beta2) of image est['img_id'] /
    est['name'] from the files other methods left in the dataset tree est['root_dir'], or from the calibration record pipe['cal_est'].
        cal_est         pickle {'sfrn': {'<camera>_<iso:05d>': (b1, b2)}, 'beta1': {camera: poly}, 'beta2': {camera: poly}}; camera and ISO
                        are fields 2 and 3 of the image name; an ISO without a record evaluates the camera's polynomials (:316-323)
        foi / liu       SIDD_Validation_Raw/{FoiEst,LiuEst}_fullPict.mat, variable 'return_params' (:324-327)
        zou             SIDD_Validation_Raw/Zou_fullPict.npy (:328-329)
        pge             SIDD_Validation_Raw/PGE_fullPict.npy; its second column is a standard deviation (:330-337)
    Returns None for the est_types that compute ('simple', 'manual').  Host-side file reads: nothing here touches the GPU."""
    est = est or {}
    est_type = str(pipe.get('est_type', 'simple'))

    def need(key):
        if est.get(key) is None:
            raise L.YondHipError(f"est_type {est_type!r} / cal_est reads a precomputed estimate: IterDenoise needs est={{'{key}': ...}} "
                                 "(YOND_SIDD.py:307-308, 316-337)")
        return est[key]
    if 'cal_est' in pipe:
        import pickle as pkl
        with open(pipe['cal_est'], 'rb') as f:
            record = pkl.load(f)
        name = need('name')
        ct, iso = name.split('_')[2], int(name.split('_')[3])
        if f'{ct}_{iso:05d}' not in record['sfrn']:
            return (np.poly1d(record['beta1'][ct])(iso), np.poly1d(record['beta2'][ct])(iso))
        r = record['sfrn'][f'{ct}_{iso:05d}']
        return (r[0], r[1])
    raw_dir = lambda: os.path.join(str(need('root_dir')), 'SIDD_Validation_Raw')
    if 'foi' in est_type or 'liu' in est_type:
        import scipy.io as sio
        fn = 'FoiEst_fullPict.mat' if 'foi' in est_type else 'LiuEst_fullPict.mat'
        r = sio.loadmat(os.path.join(raw_dir(), fn))['return_params'][need('img_id')]
        return (r[0], r[1])
    if 'zou' in est_type:
        r = np.load(os.path.join(raw_dir(), 'Zou_fullPict.npy'))[need('img_id')]
        return (r[0], r[1])
    if 'pge' in est_type:
        if est.get('est_net') is not None:
            raise NotImplementedError("est_type 'pge' with an estimation network (YOND_SIDD.py:333-335) is not built: the file form only")
        r = np.array(np.load(os.path.join(raw_dir(), 'PGE_fullPict.npy'))[need('img_id')], dtype=np.float64)
        return (r[0], r[1] ** 2)
    return None


def IterDenoise(lr_raw, net, arch, pipe, lr_full=None, p=None, device=None, log=None, biaslut=None, est=None):
    """Round 1: self-calibrated NLE -> VST -> denoise -> inverse VST; round 2 (pipe['iter']=='iter'):
    collaborative NLE from (noisy, denoised) -> guards -> second pass.  lr_raw: the SIDD layout [32][256][256]
    (denoised block by block, or -- pipe['full_dn'] -- as its 256 x 8192 concatenation, :387-389) or one Bayer
    frame [H][W] (needs pipe['full_dn']).  The collaborative estimate re-tiles into 32 vertical tiles (SIDD_256, which :431
    hard-codes) wherever the reference's split can run (packed width divisible by 32); pipe['collab_sidd256'] overrides.  Returns dict(raw_dns, regs, params) with
    device tensors in raw_dns (each [H][W], for SIDD the 256 x 8192 concatenation as in the reference).
    est: {'root_dir', 'img_id', 'name'} for the est_types that read round 1's estimate from files (file_estimate)."""
    p = dict(p or default_params())
    k = pipe.get('k', 29)
    bias_corr = pipe.get('bias_corr', 'pre')
    if bias_corr == 'none':
        bias_corr = None
    full_dn = bool(pipe.get('full_dn', False))
    if lr_full is None and 'rot_cfa' not in p:
        lr_c = _dev(lr_raw, device)
        if chain_applies(lr_c, net, arch, pipe, biaslut):
            res = _iter_denoise_chain(lr_c, net, arch, pipe, p, log=log)
            if res is not None:
                return res
    if 'rot_cfa' not in p:
        lr_c = _dev(lr_raw, device)
        lf_c = None if lr_full is None else _dev(lr_full, lr_c.device)
        if chain_applies_sidd(lr_c, lf_c, net, arch, pipe, p, biaslut):
            res = _iter_denoise_chain_sidd(lr_c, lf_c, net, arch, pipe, dict(p), log=log)
            if res is not None:
                return res
    sidd = not full_dn
    vst_type = pipe.get('vst_type', 'exact')
    scale = p['wp'] - p['bl']
    regs, params = [], []
    rot_k_ = 0
    if 'rot_cfa' in p and not full_dn:         # YOND_SIDD.py:402, 462 (the block-wise branch only)
        from .utils.sidd_utils import rot_k
        rot_k_ = rot_k(p['cfa'])
    lr = _dev(lr_raw, device)
    stack = lr.dim() == 3                      # the SIDD layout, as YOND_SIDD.eval hands it over (:507-514)
    if stack:
        if lr.shape[0] != 32:
            raise L.YondHipError("SIDD layout expects [32][256][256] blocks")
        lr_cat = torch.cat(list(lr), dim=-1).contiguous()                              # :315 (and :388 when full_dn)
        blocks = lr
    elif sidd:
        if lr.dim() != 2 or lr.shape[1] % 32:
            raise L.YondHipError("block-wise denoising (full_dn False) expects the SIDD layout [32][256][256]")
        lr_cat = lr
        blocks = torch.stack(torch.split(lr, lr.shape[1] // 32, dim=-1)).contiguous()  # :354
    else:
        lr_cat = lr
    if not pipe.get('full_est', True):         # :358-381: no estimate at all -- every block through Simple_Denoiser
        if 'pge' in str(pipe.get('est_type', '')):
            raise NotImplementedError("est_type 'pge' reads precomputed estimates from the dataset directory (:359-366)")
        if not stack:
            raise L.YondHipError("the Simple_Denoiser branch works on the SIDD stack [32][256][256]")
        outs = [Simple_Denoiser(blocks[num], net) for num in range(32)]                # :369-370
        return dict(raw_dns=[torch.cat(outs, dim=-1).contiguous()], regs=(0, 0), params=[])   # :372-378
    raw4est = lr_cat if lr_full is None else _dev(lr_full, lr.device)                  # :340
    # lr.max() for the bias LUT grid: the estimator's first kernel collects it when it reads the same frame; else a
    # reduction queued ahead of the NLE and read after the NLE's own host sync
    est_type = str(pipe.get('est_type', 'simple'))
    if 'ours' in est_type and 'cal_est' not in pipe:                                   # :342-348: NeuralNLF, a second network
        raise NotImplementedError(f"est_type {est_type!r}: the reference estimates with a second network (NeuralNLF, YOND_SIDD.py:342-348); "
                                  "this build estimates with 'simple', takes 'manual' or reads the other methods' files")
    looked_up = file_estimate(pipe, est)                                               # :316-337 (cal_est first, as the reference's chain)
    if looked_up is not None:
        reg = (np.float64(looked_up[0]), np.float64(looked_up[1]))
        lr_max_dev, nle_info = _frame_max(lr_cat), {}
    elif 'simple' in est_type:
        lr_max_dev = _frame_max(lr_cat) if lr_full is not None else None
        reg, nle_info = SimpleNLF(raw4est, k=k, setting={'mode': 'self'}, full=True)   # :341
    elif 'manual' in est_type:                                                         # :349-351: a fixed (K, sigma) = (14, 20) DN
        reg = (14 / (p['wp'] - p['bl']), (20 / (p['wp'] - p['bl'])) ** 2)
        lr_max_dev, nle_info = _frame_max(lr_cat), {}
    else:
        raise NotImplementedError(est_type)                                            # :352-353
    p['gain'], p['sigma'] = reg[0] * scale, np.sqrt(max(reg[1], 0)) * scale            # :356
    if log:
        log(f"Self Est: K={p['gain']:.4f}, b={p['sigma']:.4f} (beta1={reg[0]:.3e}, beta2={reg[1]:.3e})")
    regs.append(reg)
    params.append((p['gain'], p['sigma']))
    lr_max = np.float32(lr_max_dev.item()) if lr_max_dev is not None else np.float32(nle_info['frame_max'])


    def denoise_all(bias_func):
        if full_dn:                                                                    # :387-389
            return VST_Denoiser(lr_cat, p, net, arch, bias_corr, bias_func, vst_type, clip01=True, lr_max=lr_max, biaslut=biaslut)
        blk = blocks
        if rot_k_:                             # :402-404 / :462-464: every block turned to RGGB around the denoiser
            blk = rot90(blocks, rot_k_)
        if bias_corr is not None:
            outs = VST_Denoiser(blk, p, net, arch, bias_corr, bias_func, vst_type, clip01=True, biaslut=biaslut)   # one batch-32 forward
        else:                                  # no shared LUT: per-block calls as the reference does (:398-407)
            outs = torch.stack([VST_Denoiser(blk[num], p, net, arch, bias_corr, bias_func, vst_type, clip01=True) for num in range(32)])
        if rot_k_:
            outs = rot90(outs, 4 - rot_k_)
        return torch.cat(list(outs), dim=-1).contiguous()                              # :408

    def shared_lut():
        """:392-397 / :450-454: with the 2-D table (self.biaslut) no per-image LUT is built; outside the table get_bias is"""
        if biaslut is not None and biaslut.row(p['gain'], p['sigma'], lr.device) is not None:
            return None
        return get_bias(lr_max * scale, p['sigma'], p['gain'], device=lr.device)

    bias_func = None
    if sidd and bias_corr is not None:
        bias_func = shared_lut()                                                       # :393-395
    raw_dn = denoise_all(bias_func)
    raw_dns = [raw_dn]

    if pipe.get('iter', 'iter') == 'iter':
        for epoch in range(1, pipe.get('max_iter', 1) + 1):
            # :431 hard-codes SIDD_256=True: the packed frame is cut into 32 vertical tiles (np.split(..., 32, axis=-2), :91-93).
            # That is what runs here whenever it CAN run in the reference -- the SIDD layout, and any bare frame whose packed
            # width divides by 32 (pinned by tests/golden/iter_full.npz); for other widths (3000 x 4000: 2000 / 32) the
            # reference's split raises, and the estimate is taken without re-tiling.  pipe['collab_sidd256'] overrides.
            can_tile = (lr_cat.shape[-1] // 2) % 32 == 0
            reg = SimpleNLF(lr_cat, raw_dn, k=k,
                            setting={'mode': 'collab', 'SIDD_256': bool(pipe.get('collab_sidd256', sidd or stack or can_tile))})   # :431
            if reg[1] < 0:                                                             # :438-440
                reg = (reg[0], reg[0] ** 2)
            p['gain'], p['sigma'] = reg[0] * scale, np.sqrt(reg[1]) * scale            # :442
            if log:
                log(f"Iter {epoch} Est: K={p['gain']:.4f}, sigma={p['sigma']:.4f} (beta1={reg[0]:.3e}, beta2={reg[1]:.3e})")
            if reg[0] < 0:                                                             # :445-447
                break
            bias_func = shared_lut()                                                   # :450-454
            raw_dn = denoise_all(bias_func)
            raw_dns.append(raw_dn)
            regs.append(reg)
            params.append((p['gain'], p['sigma']))
    return dict(raw_dns=raw_dns, regs=regs, params=params)


def IterDenoiseGroup(items, net, arch, pipe, ps=None, device=None, log=None, biaslut=None, ests=None):
    """`IterDenoise` for G images of the SIDD layout at once -- items: [(lr [32][256][256], lr_full or None)], ps / ests: per-image `p` / `est`
    (or one for all).  Images are independent (YOND_SIDD.py:507-514 takes them one by one); here round 1 of the G images is ONE batch-(32 G)
    forward and round 2 another, every image with its own estimates, tables and t: per image the result is IterDenoise's (the forward bit
    for bit; the estimates to the rounding of their atomically accumulated sums, as between two IterDenoise runs).
    Images (or configurations) the grouped device chain does not cover go through IterDenoise one by one.  Returns the list of G results."""
    G = len(items)
    ps = list(ps) if isinstance(ps, (list, tuple)) else [ps] * G
    ests = list(ests) if isinstance(ests, (list, tuple)) else [ests] * G
    out = [None] * G
    pp = [dict(q or default_params()) for q in ps]
    dev_items = []
    for lr, lf in items:
        lr_c = _dev(lr, device)
        dev_items.append((lr_c, None if lf is None else _dev(lf, lr_c.device)))
    same_p = all({k: v for k, v in q.items() if k != 'cfa'} == {k: v for k, v in pp[0].items() if k != 'cfa'} for q in pp)
    if G > 1 and same_p and all(chain_applies_sidd(lr_c, lf_c, net, arch, pipe, q, biaslut) and lr_c.shape == dev_items[0][0].shape
                                 for (lr_c, lf_c), q in zip(dev_items, pp)):
        out = _iter_denoise_chain_sidd_group(dev_items, net, arch, pipe, dict(pp[0]), log=log)
    for g in range(G):
        if out[g] is None:
            out[g] = IterDenoise(items[g][0], net, arch, pipe, lr_full=items[g][1], p=ps[g], device=device, log=log, biaslut=biaslut, est=ests[g])
    return out


def denoise_stream_groups(groups, net, arch, pipe, p=None, device=None, log=None, biaslut=None, finish=None):
    """`IterDenoiseGroup` over a SEQUENCE of groups of SIDD-layout items with consecutive groups overlapped on two HIP streams (the shipped SIDD
    mode: iter, max_iter 1, block-wise denoising, est_type 'simple'): the main stream carries the batched network passes in the order D1(0), D1(1),
    D2(0), D1(2), D2(1), ...; the side stream runs group k+1's self estimates (one 16 MP frame per image) under D1(k) and group k's collaborative
    estimates + guards + parameter chains under D2(k-1).  groups: an iterable of lists [(lr [32][256][256], lr_full or None), ...]; yields one
    list of results per group, two groups late.  finish(group index, results): an optional callback run on the SIDE stream behind the group's last
    pass (the evaluation driver's block metrics: their host reads then wait for that group only, not for the passes queued since).
    Per image the result is IterDenoise's (tests/test_hip_eval.py); a group the device chain does not cover, and an image whose parameter block carries a
    flag the chain leaves to the host, go through IterDenoiseGroup / IterDenoise when they are yielded.  Items must stay unmodified until yielded."""
    p0 = dict(p or default_params())
    two = pipe.get('iter', 'iter') == 'iter' and pipe.get('max_iter', 1) == 1
    main = torch.cuda.current_stream()
    side = _side_stream(main.device)
    RING = 4
    pipe2 = dict(pipe, collab_sidd256=pipe.get('collab_sidd256', True))

    def usable(items):
        return (two and STREAM_GROUPS and len(items) >= 1 and all(chain_applies_sidd(lr, lf, net, arch, pipe, p0, biaslut) and lr.shape == items[0][0].shape
                                                                   for lr, lf in items))

    def bufs(k, g, r):
        return _chain_buffers(main.device, ('group-stream', k % RING, g, r))

    def est1(st, k, ready):
        side.wait_event(ready)
        with torch.cuda.stream(side):
            for g, (blocks, lr_full) in enumerate(st['items']):
                lr_cat = torch.cat(list(blocks), dim=-1).contiguous()                          # :315
                mx = _frame_max(lr_cat)                                                        # :393
                _chain_estimate(lr_cat if lr_full is None else lr_full, None, 'self', pipe, p0, bufs(k, g, 0), lr_max_dev=mx)
                st['lr_cat'].append(lr_cat)
                st['mx'].append(mx)
            return side.record_event()

    def est2(st, k):
        side.wait_event(st['fin1'])
        with torch.cuda.stream(side):
            for g in range(len(st['items'])):
                _chain_estimate(st['lr_cat'][g], st['out1'][g], 'collab', pipe2, p0, bufs(k, g, 1), lr_max_dev=st['mx'][g])
            return side.record_event()

    lane2 = _side_stream(main.device, 'net1') if STREAM_LANES > 1 else main     # second passes on a stream of their own: D2(k-1) beside D1(k)
    plan = _plan_of(net, main.device)

    def round_(st, k, r, wait):
        G = len(st['items'])
        lane = lane2 if r else main
        lane.wait_event(wait)
        plan.lane = 0 if lane is main else 1
        try:
            with torch.cuda.stream(lane):
                outs, watch = _chain_denoise_blocks_group([it[0] for it in st['items']], net, arch, p0, [bufs(k, g, r) for g in range(G)], 4 + 2 * (k % RING) + r)
                cat = [torch.cat(list(o), dim=-1).contiguous() for o in outs]
                fin = lane.record_event()
        finally:
            plan.lane = 0
        if lane is not main:
            for c in cat:
                c.record_stream(main)
        return cat, watch, fin

    def release(st, k):
        items = st['items']
        redone = not st['chain']
        if not st['chain']:
            res = IterDenoiseGroup(items, net, arch, pipe, ps=p0, log=log, biaslut=biaslut)
        else:
            st['fin2'].synchronize()
            trip1 = st['g1'] is not None and st['g1'].tripped()
            trip2 = st['g2'] is not None and st['g2'].tripped()
            res = []
            for g in range(len(items)):
                reg1, par1, fl1, info1 = _chain_result(bufs(k, g, 0))
                reg2, par2, fl2, info2 = _chain_result(bufs(k, g, 1))
                bad = bool(fl1 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY | PRM_BAD_ESTIMATE)) or trip1 or bool(fl2 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY))
                aborted = bool(fl2 & PRM_ROUND_ABORTED)                       # :445-447
                if not bad and not aborted:
                    bad = bool(fl2 & PRM_BAD_ESTIMATE) or trip2
                if bad:
                    res.append(IterDenoise(items[g][0], net, arch, pipe, lr_full=items[g][1], p=p0, log=log, biaslut=biaslut))
                    redone = True
                    continue
                raw_dns, regs, params = [st['out1'][g]], [reg1], [par1]
                if not aborted:
                    raw_dns.append(st['out2'][g])
                    regs.append(reg2)
                    params.append(par2)
                res.append(dict(raw_dns=raw_dns, regs=regs, params=params, nle_info=info1))
        if finish is not None:
            # on a stream of its own: the host has just waited for this group's last pass, and a host read inside `finish` must not wait for the estimates and
            # passes of the groups queued since
            aux = _side_stream(main.device, 'aux')
            if redone:
                aux.wait_stream(main)                  # (a group or an image recomputed just now, on the main stream)
            with torch.cuda.stream(aux):
                finish(k, res)
        return res

    it = iter(groups)

    def take():
        try:
            raw = next(it)
        except StopIteration:
            return None
        items = []
        for lr, lf in raw:
            lr_c = _dev(lr, device).contiguous()
            items.append((lr_c, None if lf is None else _dev(lf, lr_c.device)))
        return dict(items=items, lr_cat=[], mx=[], chain=usable(items), g1=None, g2=None)

    live = {}
    nxt = take()
    k = 0
    if nxt is not None:
        live[0] = nxt
        if nxt['chain']:
            nxt['e1'] = est1(nxt, 0, main.record_event())
    while k in live:
        st = live[k]
        nxt = take()                                  # (and mark the main stream) BEFORE queuing this group's network
        ready = main.record_event()
        if st['chain']:
            st['out1'], st['g1'], st['fin1'] = round_(st, k, 0, st['e1'])      # D1(k)
        if nxt is not None:
            live[k + 1] = nxt
            if nxt['chain']:
                nxt['e1'] = est1(nxt, k + 1, ready)                            # E1(k+1): under D1(k)
        if st['chain']:
            st['e2'] = est2(st, k)                                             # E2(k): behind D1(k), under D2(k-1)
        if k - 1 in live and live[k - 1]['chain']:
            prev = live[k - 1]
            prev['out2'], prev['g2'], prev['fin2'] = round_(prev, k - 1, 1, prev['e2'])    # D2(k-1)
        if k - 2 in live:
            yield release(live.pop(k - 2), k - 2)
        k += 1
    if k - 1 in live and live[k - 1]['chain']:
        prev = live[k - 1]
        prev['out2'], prev['g2'], prev['fin2'] = round_(prev, k - 1, 1, prev['e2'])
    for j in sorted(live):
        yield release(live.pop(j), j)
    if lane2 is not main:
        main.wait_stream(lane2)


STREAM_GROUPS = True                # (module attribute: False = the evaluation drivers take one group at a time, for A/B)


def denoise_stream_batches(frames, B, net, arch, pipe, p=None, device=None):
    """`IterDenoiseBatch` over a sequence of equally sized full Bayer frames, B per forward (BASELINE cfg 4), with the estimators of batch k+1 -- B self
    estimates and parameter chains -- on the side stream under the batched network pass of batch k (pipe['iter'] == 'once', the device chain's
    configurations; anything else: IterDenoiseBatch batch by batch).  Yields one dict per FRAME (raw_dns, regs, params), a batch late.  Per frame the
    result is IterDenoise's (tests/test_hip_eval.py).  Frames must stay unmodified until yielded."""
    p0 = dict(p or default_params())
    it = iter(frames)

    def take():
        out = []
        for f in it:
            out.append(_dev(f, device))
            if len(out) == B:
                break
        return out

    chain_cfg = (DEVICE_CHAIN and STREAM_GROUPS and pipe.get('iter', 'iter') == 'once' and pipe.get('full_dn', False) and pipe.get('bias_corr', 'pre') == 'pre'
                 and 'simple' in str(pipe.get('est_type', 'simple')) and 'cal_est' not in pipe and pipe.get('full_est', True) and 'rot_cfa' not in p0)
    if not chain_cfg:
        while True:
            batch = take()
            if not batch:
                return
            r = IterDenoiseBatch(batch, net, arch, pipe, p=p0, device=device)
            for b in range(len(batch)):
                yield dict(raw_dns=[rd[b] for rd in r['raw_dns']], regs=[rg[b] for rg in r['regs']], params=[pr[b] for pr in r['params']])
    main = torch.cuda.current_stream()
    side = _side_stream(main.device)
    NL = max(1, int(STREAM_LANES))                     # batched passes of consecutive batches alternate between NL streams (see _denoise_stream_chain)
    lanes = [main] + [_side_stream(main.device, f'net{j}') for j in range(1, NL)]
    plan = _plan_of(net, main.device)
    RING = NL + 2

    def bufs(k, b):
        return _chain_buffers(main.device, ('batch-stream', k % RING, b))

    def estimate(batch, k, ready):
        side.wait_event(ready)
        with torch.cuda.stream(side):
            for b, lr in enumerate(batch):
                _chain_estimate(lr, None, 'self', pipe, p0, bufs(k, b))
            return side.record_event()

    def release(item, k):
        batch, outs, watch, fin = item
        fin.synchronize()
        trip = watch is not None and watch.tripped()
        res = []
        for b, lr in enumerate(batch):
            reg, par, flags, info = _chain_result(bufs(k, b))
            if flags & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY | PRM_BAD_ESTIMATE) or trip:
                global DEVICE_CHAIN
                DEVICE_CHAIN = False
                try:
                    res.append(IterDenoise(lr, net, arch, pipe, p=p0))
                finally:
                    DEVICE_CHAIN = True
            else:
                res.append(dict(raw_dns=[outs[b]], regs=[reg], params=[par], nle_info=info))
        return res

    batch = take()
    if not batch:
        return
    k = 0
    est = estimate(batch, 0, main.record_event())
    pending = []
    while batch:
        nxt = take()                                   # (and mark the main stream) BEFORE queuing this batch's network
        ready = main.record_event()
        lane = lanes[k % NL]
        lane.wait_event(est)
        plan.lane = k % NL
        try:
            with torch.cuda.stream(lane):
                outs, watch = _chain_denoise_frames(batch, net, arch, p0, [bufs(k, b) for b in range(len(batch))], 4 + (k % RING))
                fin = lane.record_event()
        finally:
            plan.lane = 0
        if lane is not main:
            for o in outs:
                o.record_stream(main)
        cur = (batch, outs, watch, fin)
        if nxt:
            est = estimate(nxt, k + 1, ready)
        pending.append((cur, k))
        if len(pending) > NL:
            yield from release(*pending.pop(0))
        batch = nxt
        k += 1
    while pending:
        yield from release(*pending.pop(0))
    for ln in lanes[1:]:
        main.wait_stream(ln)


def IterDenoiseBatch(frames, net, arch, pipe, p=None, device=None):
    """`IterDenoise` for B equally sized full Bayer frames at once (BASELINE cfg 4: batch 8; pipe['full_dn']): every frame
    keeps its own noise-level estimate, bias LUT and VST constants -- the per-image steps of YOND_SIDD.py:341-356,
    392-397 -- and the B stabilised frames go through ONE batched forward per round instead of B batch-1 forwards.
    Per frame the result equals IterDenoise's.  Returns dict(raw_dns=[ [B][H][W] per round ], regs=[per round: list of B],
    params=[per round: list of B (K, sigma)], alive=[B] booleans of the frames whose round 2 ran)."""
    if not pipe.get('full_dn', False):
        raise L.YondHipError("IterDenoiseBatch handles full frames (pipe['full_dn'])")
    p0 = dict(p or default_params())
    k = pipe.get('k', 29)
    bias_corr = pipe.get('bias_corr', 'pre')
    if bias_corr == 'none':
        bias_corr = None
    vst_type = pipe.get('vst_type', 'exact')
    scale = p0['wp'] - p0['bl']
    lrs = [_dev(f, device) for f in frames]
    B = len(lrs)
    stack = torch.stack(lrs)
    est = [SimpleNLF(f, k=k, setting={'mode': 'self'}, full=True) for f in lrs]          # :341 per frame
    regs = [e[0] for e in est]
    maxes = [np.float32(e[1]['frame_max']) if 'frame_max' in e[1] else np.float32(_frame_max(f).item()) for e, f in zip(est, lrs)]
    ps = [dict(p0, gain=r[0] * scale, sigma=np.sqrt(max(r[1], 0)) * scale) for r in regs]  # :356
    raw_dn = VST_Denoiser(stack, ps, net, arch, bias_corr, None, vst_type, clip01=True, lr_max=maxes)
    raw_dns, all_regs, all_params = [raw_dn], [regs], [[(q['gain'], q['sigma']) for q in ps]]
    alive = [True] * B
    if pipe.get('iter', 'iter') == 'iter':
        for epoch in range(1, pipe.get('max_iter', 1) + 1):
            regs2, ps2, funcs = [], [], []
            for i in range(B):
                can_tile = (lrs[i].shape[-1] // 2) % 32 == 0                                # as IterDenoise: where the reference's split runs
                reg = SimpleNLF(lrs[i], raw_dn[i], k=k, setting={'mode': 'collab', 'SIDD_256': bool(pipe.get('collab_sidd256', can_tile))})
                if reg[1] < 0:                                                             # :438-440
                    reg = (reg[0], reg[0] ** 2)
                if reg[0] < 0:                                                             # :445-447: this frame keeps its round-1 result
                    alive[i] = False
                    regs2.append(reg)
                    ps2.append(ps[i])
                    funcs.append(None)
                    continue
                q = dict(p0, gain=reg[0] * scale, sigma=np.sqrt(reg[1]) * scale)           # :442
                regs2.append(reg)
                ps2.append(q)
                funcs.append(get_bias(maxes[i] * scale, q['sigma'], q['gain'], device=stack.device))   # :450-452
            if not any(alive):
                break
            nxt = VST_Denoiser(stack, ps2, net, arch, bias_corr, funcs, vst_type, clip01=True, lr_max=maxes)
            for i in range(B):
                if not alive[i]:
                    nxt[i] = raw_dn[i]
            raw_dn = nxt
            raw_dns.append(raw_dn)
            all_regs.append(regs2)
            all_params.append([(q['gain'], q['sigma']) for q in ps2])
    return dict(raw_dns=raw_dns, regs=all_regs, params=all_params, alive=alive)


def stream_applies(pipe, p=None, biaslut=None):
    """True when denoise_stream runs one of its device-chain drivers for this configuration (full-frame denoising, bias_corr 'pre' with the 1-D LUT,
    est_type 'simple', 'once' or the shipped 'iter' with max_iter 1) -- the configurations in which `frames` may carry per-frame parameter dicts."""
    ok = (DEVICE_CHAIN and bool(pipe.get('full_dn', False)) and pipe.get('bias_corr', 'pre') == 'pre' and 'simple' in str(pipe.get('est_type', 'simple'))
          and 'cal_est' not in pipe and pipe.get('full_est', True) and 'rot_cfa' not in (p or {}) and biaslut is None)
    mode = pipe.get('iter', 'iter')
    return ok and (mode == 'once' or (mode == 'iter' and pipe.get('max_iter', 1) == 1 and STREAM_ITER))


def denoise_stream(frames, net, arch, pipe, p=None, device=None):
    """`IterDenoise` over a sequence of full Bayer frames with the two phases of consecutive frames overlapped
    (pipe['iter'] == 'once', pipe['full_dn']): the noise-level estimation of frame k+1 -- memory / latency bound kernels
    and the one host round trip of the path (YOND_SIDD.py:341-356) -- runs on a second HIP stream while the
    convolution stack of frame k (:387-389) occupies the matrix cores on the first.  Same kernels, same arguments and
    therefore the same results as IterDenoise, frame by frame (tests/test_hip_pipeline.py); yields its dict per frame.
    Any other mode falls back to IterDenoise.
    Contract for device-tensor frames: frame k+1 is taken from the iterator BEFORE the network pass of frame k is queued
    and its estimate reads it on the side stream, so every frame handed in must stay UNMODIFIED until its own result has
    been yielded -- a producer that refills one buffer in place must hand in clones.
    An element of `frames` may be a pair (frame, p): a frame with its own parameter dict (wp / bl / ratio / scale: the evaluation drivers' items) --
    in the configurations `stream_applies` names."""
    chain_cfg = (DEVICE_CHAIN and pipe.get('bias_corr', 'pre') == 'pre' and 'simple' in str(pipe.get('est_type', 'simple')) and 'cal_est' not in pipe
                 and pipe.get('full_est', True) and 'rot_cfa' not in (p or {}))
    if pipe.get('full_dn', False) and pipe.get('iter', 'iter') == 'iter' and pipe.get('max_iter', 1) == 1 and chain_cfg and STREAM_ITER:
        # the reference's shipped default (runfiles/YOND/*: iter, max_iter 1; control flow YOND_SIDD.py:419-472) on the two streams
        yield from _denoise_stream_chain_iter(frames, net, arch, pipe, dict(p or default_params()), device)
        return
    if pipe.get('iter', 'iter') != 'once' or not pipe.get('full_dn', False):
        for f in frames:
            f, pk = f if isinstance(f, tuple) else (f, p)
            yield IterDenoise(f, net, arch, pipe, p=pk, device=device)
        return
    p0 = dict(p or default_params())
    if DEVICE_CHAIN and pipe.get('bias_corr', 'pre') == 'pre' and 'simple' in str(pipe.get('est_type', 'simple')) and 'cal_est' not in pipe:
        yield from _denoise_stream_chain(frames, net, arch, pipe, p0, device)
        return

    def _plain(item):
        if isinstance(item, tuple):
            raise L.YondHipError("denoise_stream: per-frame parameter dicts need a configuration of the device chain (pipeline.stream_applies)")
        return item
    frames = (_plain(f) for f in frames)
    k = pipe.get('k', 29)
    bias_corr = pipe.get('bias_corr', 'pre')
    if bias_corr == 'none':
        bias_corr = None
    vst_type = pipe.get('vst_type', 'exact')
    scale = p0['wp'] - p0['bl']
    main = torch.cuda.current_stream()
    side = _side_stream(main.device)

    def estimate(lr, ready):               # phase 1 on the side stream; returns host scalars only
        side.wait_event(ready)             # the frame as it stood when it was handed in -- NOT the work queued since
        with torch.cuda.stream(side):
            reg, info = SimpleNLF(lr, k=k, setting={'mode': 'self'}, full=True)
            lr_max = np.float32(info['frame_max'])
        return lr, reg, lr_max

    it = iter(frames)
    try:
        f = _dev(next(it), device)         # (a host array is uploaded on the main stream: the event below is behind it)
    except StopIteration:
        return
    nxt = estimate(f, main.record_event())
    pending = None                         # the previous frame's result: yielded once its range guard has been read
    k_frame = 0

    def release(item):
        res, watch = item
        if watch is not None:
            redo = watch.finish()          # the frame's forward finished a whole frame ago: no waiting in steady state
            if redo is not None:
                res['raw_dns'][0] = redo
        return res

    while nxt is not None:
        lr, reg, lr_max = nxt
        try:                               # take the next frame (and mark the main stream) BEFORE queuing this one's network
            f_next = _dev(next(it), device)
            ready = main.record_event()
        except StopIteration:
            f_next = None
        pp = dict(p0)
        pp['gain'], pp['sigma'] = reg[0] * scale, np.sqrt(max(reg[1], 0)) * scale
        raw_dn, watch = VST_Denoiser(lr, pp, net, arch, bias_corr, None, vst_type, clip01=True, lr_max=lr_max,
                                     guard='defer', guard_slot=k_frame & 1)                     # phase 2, asynchronous
        k_frame += 1
        nxt = estimate(f_next, ready) if f_next is not None else None     # overlaps with the convolutions just queued
        if pending is not None:
            yield release(pending)
        pending = (dict(raw_dns=[raw_dn], regs=[reg], params=[(pp['gain'], pp['sigma'])]), watch)
    if pending is not None:
        yield release(pending)


def _frame_item(item, p0, device):
    """An element of a stream driver's `frames`: a frame, or (frame, p) -- a frame with ITS OWN parameter dict (the evaluation drivers: wp / bl / ratio /
    scale per item, YOND_SIDD.py:503-505).  Returns (device frame, p)."""
    if isinstance(item, tuple):
        f, pk = item
        pk = dict(pk)
        if 'rot_cfa' in pk:
            raise L.YondHipError("denoise_stream: a frame with rot_cfa goes through IterDenoise (the stream drivers run the device chain)")
        return _dev(f, device), pk
    return _dev(item, device), p0


STREAM_LANES = 2                    # network passes in flight on as many HIP streams (1: every pass on the main stream, as rounds 3-5 had it)


def _denoise_stream_chain(frames, net, arch, pipe, p0, device):
    """denoise_stream on the device chain: frame k+1's estimator AND its whole parameter chain (beta -> K, sigma, t, knots ->
    bias LUT -> table) run on the side stream under the convolutions of frame k -- no host round trip anywhere; the host reads
    a frame's parameter block late, when it yields the result.  A frame whose block carries a flag (no flat area,
    table capacity, K <= 0) or whose range guard tripped is recomputed on the host-side chain before it is yielded.
    Round 6: the network passes of consecutive frames alternate between STREAM_LANES streams (each lane with its own split-plane tensors and
    FiLM vectors, engine `lane`): the persistent workgroups of frame k+1's launch take the CUs that the last, partly filled round of frame k's
    launch leaves idle (5,922 level-0 tiles = 23.1 rounds of 256) and the gap between two dependent launches of one stream is covered by the
    other's.  Same kernels, same arguments, same results; a frame is yielded STREAM_LANES frames late."""
    if STREAM_LANES > 1 and LANE_PIPELINES_ONCE:
        yield from _denoise_lanes(frames, net, arch, pipe, p0, device, 1)
        return
    main = torch.cuda.current_stream()
    side = _side_stream(main.device)
    NL = max(1, int(STREAM_LANES))
    lanes = [main] + [_side_stream(main.device, f'net{j}') for j in range(1, NL)]
    RING = NL + 2                                                # a frame's buffers rest until it has been yielded: estimate, NL passes in flight, one read
    plan = _plan_of(net, main.device)

    def estimate(item, ready, slot):
        lr, pk = item
        buf = _chain_buffers(lr.device, ('once-stream', slot % RING))
        side.wait_event(ready)                                   # the frame as it stood when it was handed in
        with torch.cuda.stream(side):
            _chain_estimate(lr, None, 'self', pipe, pk, buf)
            done = side.record_event()
        return lr, pk, buf, done

    def release(item):
        lr, pk, out, buf, watch, fin = item
        fin.synchronize()
        reg, par, flags, info = _chain_result(buf)
        if flags & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY | PRM_BAD_ESTIMATE) or (watch is not None and watch.tripped()):
            global DEVICE_CHAIN
            DEVICE_CHAIN = False
            try:
                return IterDenoise(lr, net, arch, pipe, p=pk)
            finally:
                DEVICE_CHAIN = True
        if out.device.type == 'cuda':
            out.record_stream(main)                              # (allocated on its lane's stream, handed to the caller's)
        return dict(raw_dns=[out], regs=[reg], params=[par], nle_info=info)

    it = iter(frames)
    try:
        f = _frame_item(next(it), p0, device)
    except StopIteration:
        return
    k_frame = 0
    nxt = estimate(f, main.record_event(), k_frame)
    pending = []
    try:
        while nxt is not None:
            lr, pk, buf, est_done = nxt
            try:                               # take the next frame (and mark the main stream) BEFORE queuing this one's network
                f_next = _frame_item(next(it), p0, device)
                ready = main.record_event()
            except StopIteration:
                f_next = None
            lane = lanes[k_frame % NL]
            lane.wait_event(est_done)          # (and, through the estimate's own wait, the frame as the main stream uploaded it)
            plan.lane = k_frame % NL
            with torch.cuda.stream(lane):
                out, watch = _chain_denoise(lr, net, arch, pk, buf, 4 + k_frame % RING)
                fin = lane.record_event()
            plan.lane = 0
            k_frame += 1
            nxt = estimate(f_next, ready, k_frame) if f_next is not None else None
            pending.append((lr, pk, out, buf, watch, fin))
            if len(pending) > NL:              # (NL passes stay queued behind the one the host waits for)
                yield release(pending.pop(0))
        while pending:
            yield release(pending.pop(0))
    finally:
        plan.lane = 0
        for ln in lanes[1:]:
            main.wait_stream(ln)               # (the caller's stream continues behind everything queued here)


LANE_PIPELINES_ONCE = False         # A/B: 'once' as independent lane pipelines too (no side stream)
LANE_PIPELINES = False              # 'iter' with STREAM_LANES > 1: every frame's whole chain on ONE lane (False, the default: first passes on the main stream, second passes on a lane, estimates on the side stream -- measured equal or 1 % faster)


def _denoise_lanes(frames, net, arch, pipe, p0, device, rounds):
    """STREAM_LANES independent in-order pipelines: frame k's whole chain -- E1 (self estimate + parameter chain) -> D1 (K1, network, K4) and, for
    rounds == 2 (pipe['iter'] == 'iter', max_iter 1; YOND_SIDD.py:419-472), E2 (collaborative estimate from (noisy, round-1 output) + guards + chain)
    -> D2 -- is queued on lane k mod STREAM_LANES, with no event between the lanes: one lane's estimators (latency-bound, a few workgroups) run under the
    other's network pass, and the other's launches cover a launch's last, partly filled round of persistent workgroups and the gaps between dependent
    launches.  Each lane has its own split-plane tensors and FiLM vectors (engine `lane`), each frame its own parameter blocks (a ring).  Round 2 is
    queued speculatively as in _iter_denoise_chain; a frame is yielded STREAM_LANES frames late.  Same kernels and arguments as IterDenoise."""
    main = torch.cuda.current_stream()
    NL = max(1, int(STREAM_LANES))
    lanes = [main] + [_side_stream(main.device, f'net{j}') for j in range(1, NL)]
    RING = NL + 2
    plan = _plan_of(net, main.device)

    def bufs(k, r):
        return _chain_buffers(main.device, ('lanes', k % RING, r))

    def release(st, k):
        st['fin'].synchronize()
        reg1, par1, fl1, info1 = _chain_result(bufs(k, 0))
        bad = bool(fl1 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY | PRM_BAD_ESTIMATE)) or (st['g1'] is not None and st['g1'].tripped())
        aborted = False
        if rounds == 2:
            reg2, par2, fl2, info2 = _chain_result(bufs(k, 1))
            bad = bad or bool(fl2 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY))
            aborted = bool(fl2 & PRM_ROUND_ABORTED)                           # :445-447: beta1 < 0 ends the image after round 1
            if not bad and not aborted:
                bad = bool(fl2 & PRM_BAD_ESTIMATE) or (st['g2'] is not None and st['g2'].tripped())
        if bad:                                                               # a branch the chain leaves to the host-side path
            global DEVICE_CHAIN
            DEVICE_CHAIN = False
            try:
                return IterDenoise(st['lr'], net, arch, pipe, p=st['p'])
            finally:
                DEVICE_CHAIN = True
        raw_dns, regs, params = [st['out1']], [reg1], [par1]
        if rounds == 2 and not aborted:
            raw_dns.append(st['out2'])
            regs.append(reg2)
            params.append(par2)
        return dict(raw_dns=raw_dns, regs=regs, params=params, nle_info=info1)

    pending = []
    k = 0
    try:
        for f in frames:
            lr, pk = _frame_item(f, p0, device)                               # (a host array is uploaded on the main stream)
            lane = lanes[k % NL]
            if lane is not main:
                lane.wait_event(main.record_event())                          # the frame as it stood when it was handed in
            st = dict(lr=lr, p=pk, g2=None)
            plan.lane = k % NL
            try:
                with torch.cuda.stream(lane):
                    b1 = bufs(k, 0)
                    _chain_estimate(lr, None, 'self', pipe, pk, b1)                                              # E1
                    st['out1'], st['g1'] = _chain_denoise(lr, net, arch, pk, b1, 4 + 2 * (k % RING))           # D1
                    if rounds == 2:
                        mx = torch.empty(1, dtype=torch.float32, device=main.device)
                        mx.copy_(b1.prm[PRM['frame_max']:PRM['frame_max'] + 1])                                  # float64 holding the float32 maximum -> float32
                        _chain_estimate(lr, st['out1'], 'collab', pipe, pk, bufs(k, 1), lr_max_dev=mx)           # E2
                        st['mx'] = mx
                        st['out2'], st['g2'] = _chain_denoise(lr, net, arch, pk, bufs(k, 1), 4 + 2 * (k % RING) + 1)    # D2
                    st['fin'] = lane.record_event()
            finally:
                plan.lane = 0
            if lane is not main:
                for key in ('out1', 'out2'):
                    if key in st:
                        st[key].record_stream(main)                           # (allocated on the lane's stream, handed to the caller's)
            pending.append((st, k))
            k += 1
            if len(pending) > NL:
                yield release(*pending.pop(0))
        while pending:
            yield release(*pending.pop(0))
    finally:
        plan.lane = 0
        for ln in lanes[1:]:
            main.wait_stream(ln)


STREAM_ITER = True                  # (module attribute: False = 'iter' frames one at a time, for A/B)


def _denoise_stream_chain_iter(frames, net, arch, pipe, p0, device):
    """denoise_stream for pipe['iter'] == 'iter', max_iter 1 (YOND_SIDD.py:419-472) on the device chain.  Per frame k: E1 (self estimate +
    parameter chain) -> D1 (K1, network, K4) -> E2 (collaborative estimate from (noisy, round-1 output) + guards + chain, :431-454) -> D2.
    The network passes are queued in the order D1(0), D1(1), D2(0), D1(2), D2(1), ... -- first passes on the main stream, second passes on a
    lane of their own (STREAM_LANES > 1: D2(k-1) runs beside D1(k), each covering the other's launch tails and gaps; engine `lane` 1 = its own
    split-plane tensors); the side stream runs E1(k+1) under D1(k) and E2(k) under D2(k-1) -- E2(k) needs D1(k)'s output, so frame k's second
    pass is queued behind frame k+1's first, with frame k's collaborative estimate in between on the other stream.  Round 2 is queued speculatively (whether :445-447 ends the image after round 1 is
    read from its parameter block when the frame is yielded, as in _iter_denoise_chain); the host reads a frame's blocks two frames late.
    Same kernels and arguments as IterDenoise, frame by frame: the same results (tests/test_hip_pipeline.py).  Frames handed in must stay
    unmodified until their own result has been yielded (see denoise_stream)."""
    if STREAM_LANES > 1 and LANE_PIPELINES:
        yield from _denoise_lanes(frames, net, arch, pipe, p0, device, 2)
        return
    main = torch.cuda.current_stream()
    side = _side_stream(main.device)
    RING = 4                                                                  # frames whose buffers / guard words may be live at once

    def bufs(k, r):
        return _chain_buffers(main.device, ('iter-stream', k % RING, r))

    def est1(lr, pk, ready, k):
        side.wait_event(ready)                                                # the frame as it stood when it was handed in
        with torch.cuda.stream(side):
            _chain_estimate(lr, None, 'self', pipe, pk, bufs(k, 0))
            return side.record_event()

    def est2(st, k):
        b1, b2 = bufs(k, 0), bufs(k, 1)
        side.wait_event(st['fin1'])                                           # round 1's output is complete
        with torch.cuda.stream(side):
            mx = torch.empty(1, dtype=torch.float32, device=main.device)
            mx.copy_(b1.prm[PRM['frame_max']:PRM['frame_max'] + 1])           # float64 holding the float32 maximum -> float32
            _chain_estimate(st['lr'], st['out1'], 'collab', pipe, st['p'], b2, lr_max_dev=mx)
            st['mx'] = mx
            return side.record_event()

    def release(st, k):
        st['fin2'].synchronize()
        b1, b2 = bufs(k, 0), bufs(k, 1)
        reg1, par1, fl1, info1 = _chain_result(b1)
        reg2, par2, fl2, info2 = _chain_result(b2)
        bad = bool(fl1 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY | PRM_BAD_ESTIMATE)) or (st['g1'] is not None and st['g1'].tripped())
        bad = bad or bool(fl2 & (PRM_NO_FLAT_AREA | PRM_LUT_CAPACITY))
        aborted = bool(fl2 & PRM_ROUND_ABORTED)                               # :445-447: beta1 < 0 ends the image after round 1
        if not bad and not aborted:
            bad = bool(fl2 & PRM_BAD_ESTIMATE) or (st['g2'] is not None and st['g2'].tripped())
        if bad:                                                               # a branch the chain leaves to the host-side path
            global DEVICE_CHAIN
            DEVICE_CHAIN = False
            try:
                return IterDenoise(st['lr'], net, arch, pipe, p=st['p'])
            finally:
                DEVICE_CHAIN = True
        raw_dns, regs, params = [st['out1']], [reg1], [par1]
        if not aborted:
            raw_dns.append(st['out2'])
            regs.append(reg2)
            params.append(par2)
        return dict(raw_dns=raw_dns, regs=regs, params=params, nle_info=info1)

    lane2 = _side_stream(main.device, 'net1') if STREAM_LANES > 1 else main     # second passes on a stream of their own: D2(k-1) beside D1(k)
    plan = _plan_of(net, main.device)

    def round2(st, k):
        lane2.wait_event(st['e2'])
        plan.lane = 0 if lane2 is main else 1
        try:
            with torch.cuda.stream(lane2):
                st['out2'], st['g2'] = _chain_denoise(st['lr'], net, arch, st['p'], bufs(k, 1), 4 + 2 * (k % RING) + 1)
                st['fin2'] = lane2.record_event()
        finally:
            plan.lane = 0
        if lane2 is not main:
            st['out2'].record_stream(main)                                    # (allocated on the lane's stream, handed to the caller's)

    it = iter(frames)
    try:
        f, fp = _frame_item(next(it), p0, device)
    except StopIteration:
        return
    live = {}                                                                 # frame index -> its state
    k = 0
    live[0] = dict(lr=f, p=fp)
    live[0]['e1'] = est1(f, fp, main.record_event(), 0)
    while k in live:
        st = live[k]
        try:                               # take the next frame (and mark the main stream) BEFORE queuing this one's network
            f_next, p_next = _frame_item(next(it), p0, device)
            ready = main.record_event()
        except StopIteration:
            f_next = None
        main.wait_event(st['e1'])
        st['out1'], st['g1'] = _chain_denoise(st['lr'], net, arch, st['p'], bufs(k, 0), 4 + 2 * (k % RING))       # D1(k)
        st['fin1'] = main.record_event()
        if f_next is not None:
            live[k + 1] = dict(lr=f_next, p=p_next)
            live[k + 1]['e1'] = est1(f_next, p_next, ready, k + 1)                     # E1(k+1): under D1(k)
        st['e2'] = est2(st, k)                                                # E2(k): behind D1(k), under D2(k-1)
        if k - 1 in live:
            round2(live[k - 1], k - 1)                                        # D2(k-1)
        if k - 2 in live:
            yield release(live.pop(k - 2), k - 2)
        k += 1
    if k - 1 in live:
        round2(live[k - 1], k - 1)
    for j in sorted(live):
        yield release(live.pop(j), j)
    if lane2 is not main:
        main.wait_stream(lane2)


_SIDE_STREAMS = {}


def _side_stream(dev, which='side'):
    key = (str(dev), which)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


# ------------------------------------------------------------------------------------------------
# metrics (YOND_SIDD.py:649-652, 679-697)
# ------------------------------------------------------------------------------------------------
def block_metrics(dn, hr, bh=256, bw=256):
    """Per-block PSNR (data_range 1) and SSIM (x255, 11x11 Gaussian, valid) -> two float64 arrays."""
    lib = L.load()
    dn, hr = _dev(dn), _dev(hr, dn.device if isinstance(dn, torch.Tensor) else None)
    H, W = dn.shape
    nblk = (H // bh) * (W // bw)
    nt = lib.yond_block_metrics_tiles(bh, bw)
    out = torch.empty((nblk, nt, 2), dtype=torch.float64, device=dn.device)
    L.check(lib.yond_block_metrics_f32(L.ptr(dn), L.ptr(hr), H, W, bh, bw, L.ptr(out), L.stream()), "yond_block_metrics_f32")
    s = out.sum(dim=1).cpu().numpy()
    mse = s[:, 0] / (bh * bw)
    with np.errstate(divide='ignore'):
        psnr = 10 * np.log10(1.0 / mse)
    ssim = s[:, 1] / ((bh - 10) * (bw - 10))
    return psnr, ssim
