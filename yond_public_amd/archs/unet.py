"""Denoiser plugins with the reference's registry surface (archs/Unet.py:4-104, 288-470;
archs/modules.py:163-233): classes are looked up by name (`globals()[arch['name']](arch)`,
YOND_SIDD.py:177), constructed from the runfile's `arch` dict, expose the reference's
state_dict keys / shapes (so `load_weights` / shipped checkpoints load unchanged) and are
called as `net(x)` or `net(x, t)` on NCHW float32 tensors.

The nn.Module objects here only HOLD parameters.  `forward` runs on the HIP kernels through
`engine.DenoiserPlan`; there is no torch.nn compute path and no CPU fallback: a CPU tensor or
a missing libyond_hip.so raises.
"""
import torch
import torch.nn as nn

from .. import _lib as L
from ..engine import DenoiserPlan


def _c(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2)


class _Holder(nn.Module):
    """Parameter container: children are registered under the reference's attribute names."""

    def forward(self, *a, **k):  # pragma: no cover - never used as a compute module
        raise L.YondHipError("parameter holder: compute runs in engine.DenoiserPlan")


class _GuidedBlockParams(_Holder):
    # keys: conv1, conv2, gamma.{0,2}, beta.1, short_cut.0   (archs/modules.py:163-184)
    def __init__(self, cin, c):
        super().__init__()
        self.conv1 = _c(c, c, 3)
        self.conv2 = _c(c, c, 3)
        self.gamma = nn.Sequential(_c(1, c, 1), nn.Identity(), _c(c, c, 1))
        self.beta = nn.Sequential(nn.Identity(), _c(c, c, 1))
        self.short_cut = nn.Sequential(_c(cin, c, 1)) if cin != c else nn.Sequential()


class _SNRBlockParams(_Holder):
    # keys: conv1, conv2, sfm1.{0,2}, sfm2.{0,2}, short_cut.0   (archs/modules.py:198-219)
    def __init__(self, cin, c):
        super().__init__()
        self.conv1 = _c(c, c, 3)
        self.conv2 = _c(c, c, 3)
        self.sfm1 = nn.Sequential(_c(1, c, 1), nn.Identity(), _c(c, c, 1))
        self.sfm2 = nn.Sequential(_c(1, c, 1), nn.Identity(), _c(c, c, 1))
        self.short_cut = nn.Sequential(_c(cin, c, 1)) if cin != c else nn.Sequential()


class _DownParams(_Holder):
    # key: conv   (archs/modules.py:117-125; its ReLU is dead code in the reference)
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _c(cin, cout, 3, stride=2)


class _HipDenoiser(nn.Module):
    def _common(self, args):
        self.args = args
        self.nframes = args['nframes'] if 'nframes' in args else 1
        self.cf = 0
        self.res = args['res']
        # not a reference key -- absent means 'fp32':
        #   'fp32'      fp32 results (the reference's precision): 3x3 convolutions as fp32-accurate split-operand products on
        #               the fp16 matrix cores where that kernel applies (csrc/conv_split.hip), fp32 MFMA elsewhere
        #   'fp32-mfma' every convolution on the fp32-input MFMA (Winograd / direct kernels)
        #   'fp16'      BASELINE cfg 5: operands rounded to half at the matrix core, fp32 accumulation, fp32 tensors
        self.precision = args.get('precision', 'fp32')
        if self.precision not in ('fp32', 'fp32-mfma', 'fp16'):
            raise ValueError(f"precision must be 'fp32', 'fp32-mfma' or 'fp16', got {self.precision!r}")
        self.norm = args['norm'] if 'norm' in args else False
        if args['in_nc'] * self.nframes != 4 or args['out_nc'] != 4:
            raise L.YondHipError("the HIP denoisers take packed Bayer input/output (in_nc*nframes == out_nc == 4)")
        self._plan = None
        self._plan_key = None

    def _apply(self, fn, *a, **k):
        # .to() / .float() / .cuda() replace the parameter storage: drop the packed plan and the cached list
        self._plan = None
        self._plist = None
        return super()._apply(fn, *a, **k)

    def _get_plan(self, device):
        """Packed weights + launch plan, rebuilt when a parameter was written to (in-place updates bump
        `_version`; storage replacement goes through `_apply`)."""
        plist = getattr(self, '_plist', None)
        if plist is None:
            plist = self._plist = list(self.parameters())
        key = (str(device), self.precision, tuple([p._version for p in plist]))
        if self._plan is None or self._plan_key != key:
            self._plan = DenoiserPlan(self, device)
            self._plan_key = key
        return self._plan

    def _run(self, x, t):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise L.YondHipError(f"{type(self).__name__} runs on the MI355X HIP kernels only: move the input "
                                 "(and the module) to a ROCm device; there is no CPU path")
        x = x.contiguous().float()
        with torch.no_grad():
            return self._get_plan(x.device).forward_nchw(x, t)


class GuidedResUnet(_HipDenoiser):
    """"SNR-Net" (archs/Unet.py:380-470): sigma-conditioned residual U-Net, forward(x, t)."""
    _block = _GuidedBlockParams

    def __init__(self, args=None):
        super().__init__()
        self._common(args)
        nf = args['nf']
        B = self._block
        self.conv_in = _c(4, nf, 3)
        self.conv1, self.pool1 = B(nf, nf), _DownParams(nf, nf * 2)
        self.conv2, self.pool2 = B(nf * 2, nf * 2), _DownParams(nf * 2, nf * 4)
        self.conv3, self.pool3 = B(nf * 4, nf * 4), _DownParams(nf * 4, nf * 8)
        self.conv4, self.pool4 = B(nf * 8, nf * 8), _DownParams(nf * 8, nf * 16)
        self.conv5 = B(nf * 16, nf * 16)
        self.upv6, self.conv6 = nn.ConvTranspose2d(nf * 16, nf * 8, 2, stride=2), B(nf * 16, nf * 8)
        self.upv7, self.conv7 = nn.ConvTranspose2d(nf * 8, nf * 4, 2, stride=2), B(nf * 8, nf * 4)
        self.upv8, self.conv8 = nn.ConvTranspose2d(nf * 4, nf * 2, 2, stride=2), B(nf * 4, nf * 2)
        self.upv9, self.conv9 = nn.ConvTranspose2d(nf * 2, nf, 2, stride=2), B(nf * 2, nf)
        self.conv10 = _c(nf, 4, 1)

    def forward(self, x, t):
        return self._run(x, t)


class SNRnet(GuidedResUnet):
    """archs/Unet.py:288-378: same skeleton, two multiplicative sigma gates per block."""
    _block = _SNRBlockParams


class UNetSeeInDark(_HipDenoiser):
    """archs/Unet.py:4-104: plain U-Net (LeakyReLU 0.2, 2x2 max pooling), forward(x)."""

    def __init__(self, args=None):
        super().__init__()
        self._common(args)
        nf = args['nf']
        c = 4
        for i, co in enumerate([nf, nf * 2, nf * 4, nf * 8, nf * 16], start=1):
            setattr(self, f'conv{i}_1', _c(c, co, 3))
            setattr(self, f'conv{i}_2', _c(co, co, 3))
            c = co
        for i, co in zip(range(6, 10), [nf * 8, nf * 4, nf * 2, nf]):
            setattr(self, f'upv{i}', nn.ConvTranspose2d(co * 2, co, 2, stride=2))
            setattr(self, f'conv{i}_1', _c(co * 2, co, 3))
            setattr(self, f'conv{i}_2', _c(co, co, 3))
        self.conv10_1 = _c(nf, 4, 1)

    def forward(self, x):
        return self._run(x, None)
