"""Device-side execution plan of the AWGN raw denoisers (SNR-Net = GuidedResUnet, SNRnet,
UNetSeeInDark) on the HIP kernels of libyond_hip.so.

The reference runs these nets through torch.nn / cuDNN in NCHW (archs/Unet.py:55-104, 332-378,
424-470).  Here a forward pass is a fixed sequence of launches of the MFMA implicit-GEMM kernels over
NHWC float32 activations -- by default (precision 'fp32') fp32-accurate split-operand products on the
fp16 matrix cores (csrc/conv_split.hip), with 'fp32-mfma' the fp32-input MFMA kernels (Winograd /
direct), with 'fp16' the BASELINE cfg 5 path; a range guard (DenoiserPlan.begin_guard / overflowed)
reports activations that leave fp16's range:

    [data_normalize max] -> sigma-MLPs (one launch) -> conv_in -> 9 x (conv1, conv2) with fused
    SiLU / FiLM / residual -> 4 stride-2 convs -> 4 x (convT as GEMM + pixel-shuffle store,
    two-source 1x1 shortcut: torch.cat is never materialised) -> conv_out (+x, *ub)

Weights are taken from the owning nn.Module's state_dict (reference key names / OIHW layout),
zero-padded so every channel count is a multiple of 32, re-ordered once into the LDS image order
of the kernels (`yond_pack_conv_weight_f32`) and cached on the device.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L


def _rup(c, m=32):
    return (c + m - 1) // m * m


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data)


# 3x3 stride-1 layers: 1 = Winograd F(2x2,3x3) wherever the kernel supports the shape (measured 1.45-1.5x over the direct
# kernel at 64..512 channels, 1.15x on the 32-channel level-0 layers), 0 = direct implicit GEMM everywhere
WINO_DEFAULT = 'split'              # 'split': fp32-accurate split-operand fp16-MFMA kernels where they apply, fp32 Winograd elsewhere;
                                    # 0 direct fp32 MFMA, 1 / 2 Winograd fp32 MFMA (module attribute: tools/ set it for A/B runs)


# Block-internal tensors in SPLIT PLANES (include/yond_hip.h, YondConvDesc.in_fmt / out_fmt): conv1's epilogue stores
# SiLU(FiLM(conv1)) already split into the (h, l) fp16 halves its one consumer would stage, conv2 stages by LDS-DMA alone.
# (Module attribute, not an environment switch: tools/ flip it for A/B runs.)
SPLIT_PLANES = True
SP_FLOW = True                      # ... and the whole forward in the split-plane data flow (DenoiserPlan.forward_nhwc4)
HALF_FLOW = True                    # fp16 path: the whole forward in the split-plane data flow on H-ONLY planes (2 bytes per element; round 6)
HALF_K1_TN128 = True                # ... its decoder GEMMs with >= 128-channel output pixels (and no second output) on 128-column tiles
HALF_S2_TN128 = True                # ... its stride-2 layers with >= 128 output channels (and no second output) on 8-row x 128-channel tiles
HALF_FLOW_TN128 = True              # ... its 3x3 layers with >= 128 output channels on 128-channel tiles (8 rows)
HALF_TN128 = True                   # fp16 path (precision='fp16', BASELINE cfg 5): 128-channel tiles for the 3x3 stride-1 layers with >= 128 output channels
FUSE_OUT4 = True                    # the 1x1 output projection in the epilogue of the last 3x3 convolution
FUSE_BLOCK0 = False                 # EXPERIMENT BUILDS ONLY (YOND_HIP_LIB=tools/probe/libyond_exp.so; the product library does not export the entry point):
                                    # the two level-0 residual blocks as ONE launch each (csrc/block0_fused.hip: the tensor between the convolutions stays in
                                    # LDS).  Built, parity-tested, measured (round 5): 1.16 GB less HBM traffic per block and +1.0 % per forward -- level 0 is
                                    # bound by its vector work (SiLU + split three times per block), not by bytes: profiles/r05_experiments/README.md.  Off.
K1_SUB2 = True                      # the decoder GEMMs with two sub-positions per channel tile (YondConvDesc.shuffle 2)
K1_D2_LEVELS = (3,)                 # decoder levels whose GEMM also stores SiLU(x) in split planes for the block's conv1 (YondConvDesc.dst2), as the stride-2
                                    # layers do from SP_CONV1_MIN_LEVEL down.  Measured per (GEMM + conv1) pair, two runs each (tools/layer_times.py, K1_D2_LEVELS=...):
                                    # level 3 274 -> 254-262 us, level 2 302 -> 293-303, level 1 362 -> 357-361: kept where it is clear of the noise.  () = off
K1_D2_LEVELS_HALF = ()              # ... on the fp16 path's h-only flow: none (the GEMM takes its 128-column tile instead, which has no second output)
SP_CONV1_MIN_LEVEL = 3              # from this level down a stride-2 layer also stores SiLU(x) in split planes (YondConvDesc.dst2), so that the
                                    # next block's conv1 stages by LDS-DMA alone (there conv1 would repeat the SiLU + split per output-channel tile)
UNET_SP = True                      # UNetSeeInDark: the tensor between the two convolutions of a stage in split planes (LeakyReLU applied by the producer)
SNAKE_ORDER = True                  # consecutive split-operand launches walk their tiles in opposite directions (YondConvDesc.tile_order):
                                    # a consumer starts with what its producer touched last, i.e. what the Infinity Cache still holds


def sp_plane_units(H, W):
    """YOND_SP_PLANE_UNITS: 16-byte units per plane = H*W pixels + a zero pad (conv zero padding is read from it)."""
    return (H * W + 8) // 8 * 8


class _PackedConv:
    """One convolution's device-side constants."""

    def __init__(self, dev, weight, bias, ksize, stride, splits, shuffle=False):
        """weight: OIHW float32 CPU tensor (for shuffle: ConvTranspose2d weight [Cin][Cout][2][2]);
        splits: real channel counts of the concatenated inputs (cin = sum)."""
        lib = L.load()
        w = weight.detach().to('cpu', torch.float32).numpy()
        if shuffle:
            cin, cout = w.shape[0], w.shape[1]
            # GEMM-N index n = sp*Coutp + co, sp = 2*dy+dx  <-  W[ci][co][dy][dx]
            coutp = _rup(cout)
            m = np.zeros((4, coutp, cin), np.float32)
            m[:, :cout, :] = w.transpose(2, 3, 1, 0).reshape(4, cout, cin)
            w = m.reshape(4 * coutp, cin, 1, 1)
            self.cout_real_p = coutp
            gemm_n = 4 * coutp
        else:
            cout = w.shape[0]
            coutp = _rup(cout)
            gemm_n = coutp
            self.cout_real_p = coutp
        assert sum(splits) == w.shape[1], (splits, w.shape)
        # pad every input split to a multiple of 32 (zero weights) and cout to a multiple of 32
        psplits = [_rup(s) for s in splits]
        cinp = sum(psplits)
        wp = np.zeros((gemm_n, cinp, ksize, ksize), np.float32)
        so, do = 0, 0
        for s, ps in zip(splits, psplits):
            wp[:w.shape[0], do:do + s] = w[:, so:so + s]
            so += s
            do += ps
        self._wp = np.ascontiguousarray(wp)        # padded OIHW weights (host); packed per tile width on demand
        self._packed = {}
        self._dev = dev
        b = torch.zeros(coutp, dtype=torch.float32)
        if bias is not None:
            b[:cout] = bias.detach().to('cpu', torch.float32)
        self.bias = b.to(dev)
        self.ksize, self.stride, self.shuffle = ksize, stride, shuffle
        self.psplits, self.gemm_n, self.coutp, self.cinp = psplits, gemm_n, coutp, cinp
        self.cin_real, self.cout_real = sum(splits), cout
        self.config(0, 0, 0)                       # validates the shape and packs the default layout
        self.wmax = float(np.abs(self._wp).max()) if self._wp.size else 0.0
        # algorithmic MACs per GEMM-M pixel (SURVEY.md section 8d counts real, unpadded channels)
        self.macs_per_pixel = self.cin_real * self.cout_real * (4 if shuffle else ksize * ksize)

    def config(self, N, Ho, Wo, split=False):
        """(tn, kc, packed weights on the device) for a GEMM-M extent; the packing is cached per tile width.
        split: every weight as the half pair {h, l} of the split-operand arithmetic (descriptor algo 5)."""
        lib = L.load()
        tn, kc = C.c_int(), C.c_int()
        L.check(lib.yond_conv_config(self.ksize, self.stride, self.cinp, self.gemm_n, int(self.shuffle), N, Ho, Wo,
                                     C.byref(tn), C.byref(kc)), "yond_conv_config")
        key = (tn.value, kc.value, bool(split))
        if key not in self._packed:
            packed = np.empty(self._wp.size, np.float32)
            fn = lib.yond_pack_conv_weight_split_f32 if split else lib.yond_pack_conv_weight_f32
            rc = fn(_np_ptr(self._wp), self.gemm_n, self.cinp, self.ksize, tn.value, kc.value, _np_ptr(packed))
            if split and rc == -2:          # a weight outside fp16's range: the caller takes the fp32 packing
                self._packed[key] = None
            else:
                L.check(rc, "yond_pack_conv_weight_f32")
                self._packed[key] = torch.from_numpy(packed).to(self._dev)
        return tn.value, kc.value, self._packed[key]

    def wino(self):
        """(tn, packed Winograd F(2x2,3x3) weights) or None when the layer does not fit that kernel."""
        lib = L.load()
        if self.ksize != 3 or self.stride != 1 or self.shuffle:
            return None
        tn = int(lib.yond_conv_wino_supported(self.cinp, self.gemm_n))
        if not tn:
            return None
        if 'wino' not in self._packed:
            packed = np.empty(16 * self.gemm_n * self.cinp, np.float32)
            L.check(lib.yond_pack_conv_wino_weight_f32(_np_ptr(self._wp), self.gemm_n, self.cinp, tn, _np_ptr(packed)),
                    "yond_pack_conv_wino_weight_f32")
            self._packed['wino'] = (tn, torch.from_numpy(packed).to(self._dev))
        return self._packed['wino']

    def split(self, parts=2, wide=True, wide_s2=False):
        """(tn, weights packed for the split-operand fp16-MFMA kernel) or None when the layer does not fit it.
        wide: h-only operands of a plain-tensor 3x3 layer may take 128-channel tiles (the split-plane flow's kernels do not)."""
        lib = L.load()
        if self.shuffle != (self.ksize == 1):
            return None                     # ksize 1: the decoder's pixel-shuffle GEMM only (48-channel steps, >= 64 channels)
        tn = int(lib.yond_conv_split_supported(self.ksize, self.stride, self.cinp, self.gemm_n))
        if not tn:
            return None
        if parts == 1 and wide and HALF_TN128 and tn == 64 and self.ksize == 3 and self.stride == 1 and self.gemm_n % 128 == 0:
            tn = 128                        # h-only operands: one accumulator per block leaves room for two blocks per wave (conv_split_kernel.h)
        if parts == 1 and wide_s2 and tn == 64 and self.ksize == 3 and self.stride == 2 and self.gemm_n % 128 == 0:
            tn = 128                        # ... the flow's stride-2 layers likewise (8-row tiles, two rows x two channel blocks per wave)
        if parts == 1 and wide_s2 and tn == 64 and self.ksize == 1 and self.shuffle == 1 and (self.gemm_n // 4) % 128 == 0:
            tn = 128                        # ... and its decoder GEMMs whose output pixels have >= 128 channels
        key = ('split', parts) if tn != 128 else ('split', parts, 128)     # (train.py refreshes ('split', 2) in place every step)
        if key not in self._packed:
            packed = np.empty(self._wp.size * parts // 2, np.float32)
            rc = lib.yond_pack_conv_split_weight_f32(_np_ptr(self._wp), self.gemm_n, self.cinp, self.ksize, tn, parts, _np_ptr(packed))
            if rc == -2:                    # a weight outside fp16's range: this layer stays on the fp32-input MFMA kernels
                self._packed[key] = None
                return None
            L.check(rc, "yond_pack_conv_split_weight_f32")
            self._packed[key] = (tn, torch.from_numpy(packed).to(self._dev))
        return self._packed[key]


class _PackedUpSub2:
    """A decoder GEMM in the two-sub-positions-per-tile form (YondConvDesc.shuffle 2, conv_split_kernel.h S2):
    K = [cur 2c | skip at dx = 0: c | skip at dx = 1: c | zero columns up to a multiple of 48]; GEMM columns ordered
    [dy][channel block of 32][dx][32]; the skip weights of sub-position (dy, dx) sit in the dx range, the other range is zero."""

    def __init__(self, dev, w_f, b_f, c):
        """w_f: the folded ConvTranspose2d-layout weights [2c (cur) + c (skip)][c][2][2] of DenoiserPlan; b_f [c]."""
        lib = L.load()
        assert c % 32 == 0 and w_f.shape == (3 * c, c, 2, 2)
        w = w_f.detach().to('cpu', torch.float32).numpy()
        k = (4 * c + 47) // 48 * 48
        m = np.zeros((2, c // 32, 2, 32, k), np.float32)               # [dy][channel block][dx][channel % 32][k]
        for dy in range(2):
            for dx in range(2):
                wc = w[:2 * c, :, dy, dx].T.reshape(c // 32, 32, 2 * c)             # [block][co % 32][cur channel]
                ws = w[2 * c:, :, dy, dx].T.reshape(c // 32, 32, c)
                m[dy, :, dx, :, :2 * c] = wc
                m[dy, :, dx, :, 2 * c + dx * c:2 * c + (dx + 1) * c] = ws
        self._wp = np.ascontiguousarray(m.reshape(4 * c, k, 1, 1))
        self._dev, self._wpk = dev, {}
        self.ok = self._pack(2) is not None                            # (-2: a weight outside fp16's range -> the plain form)
        b = torch.zeros(c, dtype=torch.float32)
        b[:] = b_f.detach().to('cpu', torch.float32)
        self.bias = b.to(dev)
        self.psplits, self.gemm_n, self.coutp, self.cinp = [2 * c, k - 2 * c], 4 * c, c, k
        self.ksize, self.stride, self.shuffle = 1, 1, 2
        self.cout_real_p = c
        self.wmax = float(np.abs(self._wp).max())
        self.macs_per_pixel = 2 * c * c * 4 + 2 * c * c * 4

    def _pack(self, parts):
        if parts not in self._wpk:
            packed = np.empty(self._wp.size * parts // 2, np.float32)
            rc = L.load().yond_pack_conv_split_weight_f32(_np_ptr(self._wp), self._wp.shape[0], self._wp.shape[1], 1, 64, parts, _np_ptr(packed))
            self._wpk[parts] = torch.from_numpy(packed).to(self._dev) if rc == 0 else None
        return self._wpk[parts]

    def split(self, parts=2, wide=True, wide_s2=False):
        return (64, self._pack(parts)) if (self.ok and self._pack(parts) is not None) else None

    def wino(self):
        return None


class DenoiserPlan:
    """Packs a module's parameters once and runs forwards on NHWC4 device tensors."""

    def __init__(self, module, device):
        self.lib = L.load()
        self.dev = torch.device(device)
        self.prof = None                           # list -> record (kernel tag, flops, start, end) HIP events per conv launch
        self.status = torch.zeros(12, dtype=torch.int32, device=self.dev)    # range-guard words of the half-precision paths (0-2: pipeline wrappers, 3: the
                                                                             # plugin surface, 4-11: the two-stream 'iter' driver's ring)
        self.status_slot = 0
        self.strict = False                        # True: every convolution on the fp32-input MFMA kernels (guard fallback)
        self.precision = getattr(module, 'precision', 'fp32')      # 'fp16': MFMA convolutions on the fp16 matrix path (cfg 5)
        self.kind = type(module).__name__          # GuidedResUnet | SNRnet | UNetSeeInDark
        self.res = bool(module.res)
        self.norm = bool(module.norm)
        sd = {k: v.detach() for k, v in module.state_dict().items()}
        self.sd = sd
        self.guided = self.kind in ('GuidedResUnet', 'SNRnet')
        dev = self.dev
        if self.guided:
            nf = sd['conv_in.weight'].shape[0]
            self.nf = nf
            self.conv_in_w, self.conv_in_b = self._pack_conv_in(sd['conv_in.weight'], sd['conv_in.bias'])
            self.blocks = {}
            chans = [nf, nf * 2, nf * 4, nf * 8, nf * 16, nf * 8, nf * 4, nf * 2, nf]
            film = []
            for i, c in enumerate(chans, start=1):
                pre = f'conv{i}'
                blk = {'C': c, 'Cp': _rup(c)}
                blk['conv1'] = _PackedConv(dev, sd[pre + '.conv1.weight'], None, 3, 1, [c])
                blk['conv2'] = _PackedConv(dev, sd[pre + '.conv2.weight'], None, 3, 1, [c])
                if i >= 6:
                    # Decoder: up = ConvTranspose2d(cur); x = short_cut(cat(up, skip)).  `up` feeds nothing else
                    # (archs/Unet.py:451-466, modules.py:186-187), and both maps are linear per sub-position, so they
                    # fold into ONE GEMM per sub-position: x = (Wsc_up . Wt[dy,dx]) cur + Wsc_skip skip + (bsc + Wsc_up bt).
                    # The product is formed once in float64; `up` (the largest tensor of the level) is never written.
                    wt = sd[f'upv{i}.weight'].to('cpu', torch.float64)                   # [2c][c][2][2]
                    bt = sd[f'upv{i}.bias'].to('cpu', torch.float64)
                    wsc = sd[pre + '.short_cut.0.weight'].to('cpu', torch.float64)[:, :, 0, 0]   # [c][up c | skip c]
                    bsc = sd[pre + '.short_cut.0.bias'].to('cpu', torch.float64)
                    w_cur = torch.einsum('ou,iuyx->ioyx', wsc[:, :c], wt)                # [2c][c][2][2]
                    w_skip = wsc[:, c:].t()[:, :, None, None].expand(c, c, 2, 2)        # [c (skip in)][c][2][2]
                    w_f = torch.cat([w_cur, w_skip], 0).to(torch.float32).contiguous()   # ConvTranspose2d layout over [cur | skip]
                    b_f = (bsc + wsc[:, :c] @ bt).to(torch.float32)
                    blk['upsc'] = _PackedConv(dev, w_f, b_f, 1, 1, [2 * c, c], shuffle=True)
                    # algorithmic MACs of the two reference layers, per GEMM-M (low resolution) pixel
                    blk['upsc'].macs_per_pixel = 2 * c * c * 4 + 2 * c * c * 4
                    if c == 32:                                      # (measured: 326 -> 256 us at 32 channels; at 64 / 128 / 256 the 1.33-1.5x
                        up2 = _PackedUpSub2(dev, w_f, b_f, c)        #  longer K costs more than the shared staging saves: +35 ... +50 us)
                        if up2.ok:
                            blk['upsc2'] = up2
                if i <= 4:
                    blk['pool'] = _PackedConv(dev, sd[f'pool{i}.conv.weight'], sd[f'pool{i}.conv.bias'], 3, 2, [c])
                self.blocks[i] = blk
                film.append((pre, c))
            self._film_spec = film
            self._film_cache = {}
            self.w_out = self._pad_out_w(sd['conv10.weight'])
            self.b_out = sd['conv10.bias'].to(dev, torch.float32).contiguous()
            self._film_params = {k: v.to(dev, torch.float32).contiguous() for k, v in sd.items()
                                 if any(s in k for s in ('.gamma.', '.beta.', '.sfm1.', '.sfm2.', '.conv1.bias', '.conv2.bias'))}
        else:
            nf = sd['conv1_1.weight'].shape[0]
            self.nf = nf
            self.conv_in_w, self.conv_in_b = self._pack_conv_in(sd['conv1_1.weight'], sd['conv1_1.bias'])
            self.convs = {}
            c = nf
            self.convs['conv1_2'] = _PackedConv(dev, sd['conv1_2.weight'], sd['conv1_2.bias'], 3, 1, [c])
            for i in range(2, 6):
                self.convs[f'conv{i}_1'] = _PackedConv(dev, sd[f'conv{i}_1.weight'], sd[f'conv{i}_1.bias'], 3, 1, [c])
                c *= 2
                self.convs[f'conv{i}_2'] = _PackedConv(dev, sd[f'conv{i}_2.weight'], sd[f'conv{i}_2.bias'], 3, 1, [c])
            for i in range(6, 10):
                self.convs[f'upv{i}'] = _PackedConv(dev, sd[f'upv{i}.weight'], sd[f'upv{i}.bias'], 1, 1, [c], shuffle=True)
                c //= 2
                self.convs[f'conv{i}_1'] = _PackedConv(dev, sd[f'conv{i}_1.weight'], sd[f'conv{i}_1.bias'], 3, 1, [c, c])
                self.convs[f'conv{i}_2'] = _PackedConv(dev, sd[f'conv{i}_2.weight'], sd[f'conv{i}_2.bias'], 3, 1, [c])
            self.w_out = self._pad_out_w(sd['conv10_1.weight'])
            self.b_out = sd['conv10_1.bias'].to(dev, torch.float32).contiguous()

    # -- packing helpers -------------------------------------------------------------------
    def _pack_conv_in(self, w, b):
        w = w.detach().to('cpu', torch.float32).numpy()
        cout = w.shape[0]
        if w.shape[1] != 4:
            raise L.YondHipError(f"first layer must have 4 input channels (packed Bayer), got {w.shape[1]}")
        coutp = _rup(cout)
        wp = np.zeros((coutp, 4, 3, 3), np.float32)
        wp[:cout] = w
        packed = np.empty(coutp * 40, np.float32)
        L.check(self.lib.yond_pack_conv_in_weight_f32(_np_ptr(wp), coutp, _np_ptr(packed)), "yond_pack_conv_in_weight_f32")
        bp = torch.zeros(coutp, dtype=torch.float32)
        bp[:cout] = b.detach().to('cpu', torch.float32)
        return torch.from_numpy(packed).to(self.dev), bp.to(self.dev)

    def _pad_out_w(self, w):
        w = w.detach().to('cpu', torch.float32).reshape(w.shape[0], -1)
        if w.shape[0] != 4:
            raise L.YondHipError(f"last layer must have 4 output channels, got {w.shape[0]}")
        wp = torch.zeros(4, _rup(w.shape[1]), dtype=torch.float32)
        wp[:, :w.shape[1]] = w
        return wp.to(self.dev).contiguous()

    # -- launches ----------------------------------------------------------------------------
    def _split_pair(self, pc1, pc2):
        """Both 3x3 layers of a residual block run on the split-operand kernel (algo 3, or 4 on the fp16 path): the SiLU
        between them may then sit in the producer's epilogue (descriptor post_act 1)."""
        if getattr(self, 'strict', False) or getattr(self, 'conv_algo', WINO_DEFAULT) != 'split':
            return False
        prec = getattr(self, 'precision', 'fp32')
        if prec not in ('fp32', 'fp16'):
            return False
        parts = 2 if prec == 'fp32' else 1
        return all(pc.ksize == 3 and pc.stride == 1 and pc.split(parts) is not None for pc in (pc1, pc2))

    def _sp_flow(self, N, H, W):
        """True when a guided net's whole forward runs in the split-plane data flow (see forward_nhwc4): fp32 results at
        split precision, every 3x3 / stride-2 / decoder layer on the split-operand kernel, the output projection fused."""
        if not (SPLIT_PLANES and SP_FLOW) or getattr(self, 'strict', False) or getattr(self, 'conv_algo', WINO_DEFAULT) != 'split':
            return False
        prec = getattr(self, 'precision', 'fp32')
        if prec not in ('fp32', 'fp16') or (prec == 'fp16' and not HALF_FLOW) or sp_plane_units(H, W) * 64 * 4 >= 2 ** 31:
            return False
        parts = 2 if prec == 'fp32' else 1               # fp16 path (BASELINE cfg 5): the same flow on h-only planes, 2 bytes per element
        for i, blk in self.blocks.items():
            for k in ('conv1', 'conv2', 'pool', 'upsc'):
                if k in blk and blk[k].split(parts, wide=False) is None:
                    return False
        return self._out4_fusable(self.blocks[9]['conv2'], flow=True)

    def _out4_fusable(self, pc, flow=False):
        """The 1x1 output projection can ride in the epilogue of the last 3x3 convolution: split kernel, one 32-channel tile (on the fp16 path
        only inside the split-plane flow: its h-only kernel with the projection takes h-only planes)."""
        prec = getattr(self, 'precision', 'fp32')
        return (not getattr(self, 'strict', False) and getattr(self, 'conv_algo', WINO_DEFAULT) == 'split' and (prec == 'fp32' or (prec == 'fp16' and flow))
                and pc.ksize == 3 and pc.stride == 1 and pc.gemm_n == 32 and pc.split(2 if prec == 'fp32' else 1) is not None and FUSE_OUT4)

    def _new_sp(self, key, N, H, W, Cc, parts=2):
        """A split-plane tensor [N][Cc/16][2][parts][sp_plane_units(H, W)] x 16 bytes, kept per (key, shape) across forwards: the
        zero units behind every plane are written once, here (producers never touch them).  parts 1: h-only planes (the fp16 path)."""
        cache = self.__dict__.setdefault('_sp_cache', {})
        k = (key, N, H, W, Cc, parts, getattr(self, 'lane', 0))              # (lane: the forwards of two frames in flight on two streams keep their own tensors)
        if k not in cache:
            cache[k] = torch.zeros(N * (Cc // 16) * 2 * parts * sp_plane_units(H, W) * 4, dtype=torch.float32, device=self.dev)
        return cache[k]

    def _conv(self, pc, src0, src1, N, H, W, dst, escale=None, eshift=None, ebatch=0, res=None, pre_act=0, post_act=0,
              slope=0.0, algo=None, out4=None, in_fmt=0, out_fmt=0, res_fmt=0, dst2=None):
        d = L.YondConvDesc()
        d.src0 = src0.data_ptr()
        d.src1 = src1.data_ptr() if src1 is not None else None
        d.C0 = pc.psplits[0]
        d.C1 = pc.psplits[1] if len(pc.psplits) > 1 else 0
        d.N, d.H, d.W = N, H, W
        if pc.stride == 2:
            d.Ho, d.Wo = (H + 1) // 2, (W + 1) // 2
        else:
            d.Ho, d.Wo = H, W
        d.Cout = pc.gemm_n
        d.ksize, d.stride, d.shuffle = pc.ksize, pc.stride, int(pc.shuffle)
        d.pre_act, d.post_act, d.slope = pre_act, post_act, slope
        # `algo` (tests) / plan.conv_algo: 0 direct fp32, 1 Winograd fp32 where it is the faster kernel, 2 Winograd
        # fp32 wherever supported; plan.precision 'fp16' (BASELINE cfg 5): every MFMA convolution on the fp16 matrix path
        prec = getattr(self, 'precision', 'fp32')
        if getattr(self, 'strict', False) and algo is None:
            prec, algo = 'fp32-mfma', 1
        if algo is None:
            algo = getattr(self, 'conv_algo', WINO_DEFAULT)
            if prec == 'fp32-mfma' and algo == 'split':
                algo = 1
            if prec == 'fp16' and WINO_DEFAULT == 'split':
                algo = 'half'                    # 3x3: h halves staged once (conv_split.hip); other layers: algo 2 below
        fp16 = prec == 'fp16' or algo == 'fp16'
        wino = split = None
        if algo in ('split', 'half'):
            # (h-only operands: 128-channel tiles for plain tensors only; the decoder GEMM only inside the split-plane flow)
            # (the stride-2 layers of the h-only flow: 128-channel tiles where 8-row tiles fill the 256 workgroups and there is no second output --
            #  that instantiation spills)
            ws2 = (HALF_S2_TN128 and algo == 'half' and pc.stride == 2 and in_fmt == 1 and out_fmt == 2 and dst2 is None
                   and (pc.gemm_n // 128) * ((d.Wo + 31) // 32) * ((d.Ho + 7) // 8) * N >= 256)
            ws2 = ws2 or (HALF_K1_TN128 and algo == 'half' and pc.ksize == 1 and in_fmt == 1 and dst2 is None)
            split = pc.split(2 if algo == 'split' else 1, wide=(not (in_fmt or out_fmt or res_fmt or dst2 is not None)) or (HALF_FLOW_TN128 and out_fmt == 1),
                             wide_s2=ws2)
            if algo == 'half' and pc.ksize == 1 and in_fmt != 1:
                split = None
            if split is None and not fp16 and pc.ksize == 3:
                wino = pc.wino()                     # 3x3 layers the split kernel does not take
        elif not fp16:
            wino = pc.wino() if algo in (1, 2) else None
        if split is not None:
            tn, wpk = split
            kc = 16
            d.algo = 3 if algo == 'split' else 4
        elif wino is not None:
            tn, wpk = wino
            kc = 8
            d.algo = 1
        elif algo == 'split' and pc.ksize == 1 and not fp16 and pc.config(N, d.Ho, d.Wo, split=True)[2] is not None:
            tn, kc, wpk = pc.config(N, d.Ho, d.Wo, split=True)       # 1x1 / transposed layers: split operands in the generic kernel
            d.algo = 5
        else:
            tn, kc, wpk = pc.config(N, d.Ho, d.Wo)
            d.algo = 2 if (fp16 and pc.wmax <= 65504.0) else 0       # (a weight outside fp16's range keeps the layer in fp32)
        d.wpk = wpk.data_ptr()
        d.tn = tn
        d.kc = kc
        d.escale = escale.data_ptr() if escale is not None else None
        d.eshift = (eshift if eshift is not None else pc.bias).data_ptr()
        d.ebatch = ebatch
        d.res = res.data_ptr() if res is not None else None
        d.dst = dst.data_ptr() if dst is not None else None
        status = getattr(self, 'status', None)                       # (bare plans of the kernel tests have none)
        d.status = status.data_ptr() + 4 * self.status_slot if status is not None else None
        d.in_fmt, d.out_fmt, d.res_fmt = in_fmt, out_fmt, res_fmt
        d.dst2 = dst2.data_ptr() if dst2 is not None else None
        if SNAKE_ORDER and d.algo in (3, 4):
            self._tile_order = 1 - getattr(self, '_tile_order', 0)      # (the first layer, conv_in, runs first row to last)
            d.tile_order = self._tile_order
        clk = getattr(self, 'clk', None)             # bench.py: in-kernel clock of the split-operand launches (int64[2] on the device)
        d.clk = clk.data_ptr() if (clk is not None and d.algo in (3, 4)) else None
        if (in_fmt or out_fmt or res_fmt) and d.algo not in (3, 4):
            raise L.YondHipError("split-plane / 4-channel-plane tensors need the split-operand kernel (algo 3, or 4 with h-only planes)")
        if out4 is not None:
            # (w [4][Cout], bias [4], network input NHWC4 or None, per-image maxima or None, destination NHWC4)
            w4, b4, x4, ub4, o4 = out4
            if d.algo not in (3, 4):
                raise L.YondHipError("fused output projection needs the split-operand 3x3 kernel")
            d.out4_w, d.out4_b = w4.data_ptr(), (b4.data_ptr() if b4 is not None else None)
            d.out4_x = x4.data_ptr() if x4 is not None else None
            d.out4_ub = ub4.data_ptr() if ub4 is not None else None
            d.out4_dst = o4.data_ptr()
        prof = getattr(self, 'prof', None)
        if prof is not None:
            tag = f"conv_wino_kernel<{tn}>" if d.algo == 1 else f"conv_mfma_kernel<{pc.ksize},{pc.stride},8,{tn},{kc}>"
            if d.algo in (3, 4):
                tag = f"conv_split_kernel<{pc.stride if pc.ksize == 3 else 'k1'},{tn},{5 - d.algo}>"
            if d.algo == 2:
                tag += "/f16"
            if d.algo == 5:
                tag += "/split"
            only = getattr(self, 'prof_only', None)      # bench.py: events around one kernel family only (an event pair
            if only is not None and not only(tag):       # costs the stream a few microseconds)
                prof = None
            every = getattr(self, 'prof_every', 1)       # ... and only in every n-th forward
            if every > 1 and getattr(self, '_fwd_idx', 0) % every:
                prof = None
        if prof is not None:
            # events on the stream the kernel is launched on (torch's current stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(self.lib.yond_conv2d_f32(C.byref(d), L.stream()), "yond_conv2d_f32")
        if prof is not None:
            e1.record()
            prof.append((tag, 2.0 * pc.macs_per_pixel * N * d.Ho * d.Wo, e0, e1))
        return dst

    def _block0_weights(self, pc):
        """A 32 -> 32 3x3 layer's weights in the fused level-0 kernel's LDS order (yond_pack_block0_weight_f32), cached on the layer."""
        w = getattr(pc, '_b0', None)
        if w is None:
            if not L.has("yond_pack_block0_weight_f32"):
                raise L.YondHipError("the fused level-0 block exists only in experiment builds of the library (python -m yond_public_amd.build --experiments)")
            if not (pc.ksize == 3 and pc.stride == 1 and pc.gemm_n == 32 and pc.cinp == 32 and len(pc.psplits) == 1):
                return None
            host = np.zeros(9 * 2 * 4 * 32 * 8, np.float16)
            if self.lib.yond_pack_block0_weight_f32(_np_ptr(pc._wp), 32, 32, _np_ptr(host)) != 0:
                return None                                  # (a weight beyond fp16's range: the two-launch path's packers decide)
            w = pc._b0 = torch.from_numpy(host.view(np.float32)).to(pc._dev)
        return w

    def _block0(self, pc1, pc2, x, N, H, W, f, in_fmt, dst=None, out4=None):
        """One level-0 residual block in ONE launch (csrc/block0_fused.hip): out = conv2(SiLU(conv1(SiLU(x)) * f0 + f1)) * f2 + f3 + x.
        x: planes of 4 channels (in_fmt 2) or [N][H][W][32] (0); dst: split planes, or out4 = (w4, b4, x4, ub, destination NHWC4)."""
        w1, w2 = self._block0_weights(pc1), self._block0_weights(pc2)
        d = L.YondBlock0Desc()
        d.x, d.in_fmt, d.N, d.H, d.W = x.data_ptr(), in_fmt, N, H, W
        d.w1, d.w2 = w1.data_ptr(), w2.data_ptr()
        d.s1, d.t1, d.s2, d.t2 = (None if v is None else v.data_ptr() for v in f)
        d.ebatch = 1
        d.dst = dst.data_ptr() if dst is not None else None
        if out4 is not None:
            w4, b4, x4, ub4, o4 = out4
            d.out4_w, d.out4_b = w4.data_ptr(), (b4.data_ptr() if b4 is not None else None)
            d.out4_x = x4.data_ptr() if x4 is not None else None
            d.out4_ub = ub4.data_ptr() if ub4 is not None else None
            d.out4_dst = o4.data_ptr()
        status = getattr(self, 'status', None)
        d.status = status.data_ptr() + 4 * self.status_slot if status is not None else None
        prof = getattr(self, 'prof', None)
        if prof is not None:
            tag = "block0_fused_kernel" + ("<o4>" if out4 is not None else "")
            only = getattr(self, 'prof_only', None)
            if (only is not None and not only(tag)) or (getattr(self, 'prof_every', 1) > 1 and getattr(self, '_fwd_idx', 0) % getattr(self, 'prof_every', 1)):
                prof = None
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(self.lib.yond_block0_fused_f32(C.byref(d), L.stream()), "yond_block0_fused_f32")
        if prof is not None:
            e1.record()
            prof.append((tag, 2.0 * (pc1.macs_per_pixel + pc2.macs_per_pixel) * N * H * W, e0, e1))
        return dst

    def _film(self, t_dev, ub, N):
        """All nine blocks' (scale, shift) epilogue vectors in one launch."""
        key = (N, getattr(self, 'lane', 0))
        if key not in self._film_cache:
            fp = self._film_params
            outs, descs = {}, (L.YondFilmDesc * len(self._film_spec))()
            for bi, (pre, c) in enumerate(self._film_spec):
                cp = _rup(c)
                o = torch.zeros(4, N, cp, dtype=torch.float32, device=self.dev)
                outs[pre] = o
                d = descs[bi]
                d.C, d.ld = c, cp
                if self.kind == 'GuidedResUnet':
                    d.kind = 0
                    d.w_a0, d.b_a0 = fp[pre + '.gamma.0.weight'].data_ptr(), fp[pre + '.gamma.0.bias'].data_ptr()
                    d.w_a2, d.b_a2 = fp[pre + '.gamma.2.weight'].data_ptr(), fp[pre + '.gamma.2.bias'].data_ptr()
                    d.w_b0 = d.b_b0 = None
                    d.w_b, d.b_b = fp[pre + '.beta.1.weight'].data_ptr(), fp[pre + '.beta.1.bias'].data_ptr()
                else:
                    d.kind = 1
                    d.w_a0, d.b_a0 = fp[pre + '.sfm1.0.weight'].data_ptr(), fp[pre + '.sfm1.0.bias'].data_ptr()
                    d.w_a2, d.b_a2 = fp[pre + '.sfm1.2.weight'].data_ptr(), fp[pre + '.sfm1.2.bias'].data_ptr()
                    d.w_b0, d.b_b0 = fp[pre + '.sfm2.0.weight'].data_ptr(), fp[pre + '.sfm2.0.bias'].data_ptr()
                    d.w_b, d.b_b = fp[pre + '.sfm2.2.weight'].data_ptr(), fp[pre + '.sfm2.2.bias'].data_ptr()
                d.cb1, d.cb2 = fp[pre + '.conv1.bias'].data_ptr(), fp[pre + '.conv2.bias'].data_ptr()
                d.s1, d.t1, d.s2, d.t2 = (o[j].data_ptr() for j in range(4))
            raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.dev)
            self._film_cache[key] = (outs, raw, len(self._film_spec))
        outs, raw, nb = self._film_cache[key]
        L.check(self.lib.yond_film_f32(L.ptr(raw), nb, L.ptr(t_dev), L.ptr(ub), N, L.stream()), "yond_film_f32")
        return outs

    # -- range guard of the half-precision operand paths -------------------------------------------------------
    def uses_half_operands(self):
        """True when forwards stage activations as fp16 (split operands or the fp16 path): they need |a| <= 65504."""
        return not self.strict and getattr(self, 'precision', 'fp32') in ('fp32', 'fp16') and getattr(self, 'conv_algo', WINO_DEFAULT) in ('split', 'half', 'fp16')

    def begin_guard(self, slot=0):
        """Zero the status word `slot` and direct the following launches to it (asynchronous)."""
        self.status_slot = slot
        self.status[slot:slot + 1].zero_()

    def overflowed(self, slot=0):
        """Read the status word (synchronises the current stream): True if a staged activation left fp16's range."""
        return bool(int(self.status[slot].item()) & 1)

    def image_max(self, x4, N):
        elems = x4.numel() // N
        partial = torch.empty(N * 256, dtype=torch.float32, device=self.dev)
        ub = torch.empty(N, dtype=torch.float32, device=self.dev)
        L.check(self.lib.yond_image_max_f32(L.ptr(x4), N, elems, L.ptr(partial), L.ptr(ub), L.stream()), "yond_image_max_f32")
        return ub

    def _new(self, N, H, W, Cc):
        return torch.empty((N, H, W, Cc), dtype=torch.float32, device=self.dev)

    def forward_nhwc4(self, x4, t_dev=None, ub=None):
        """x4: [N][H][W][4] float32 device tensor (H, W multiples of 16); t_dev: [N] float32 (guided
        nets); ub: optional precomputed per-image maximum [N] (K1 provides it).  Returns [N][H][W][4]."""
        L.require_cuda(x4, "x")
        self._fwd_idx = getattr(self, '_fwd_idx', -1) + 1
        self._tile_order = 0
        N, H, W, c4 = x4.shape
        if c4 != 4 or H % 16 or W % 16:
            raise L.YondHipError(f"input must be [N][H][W][4] with H, W multiples of 16, got {tuple(x4.shape)}")
        if self.norm and ub is None:
            ub = self.image_max(x4, N)
        if not self.norm:
            ub = None
        st = L.stream()
        lib = self.lib
        nfp = _rup(self.nf)
        if self.guided:
            if t_dev is None:
                raise L.YondHipError(f"{self.kind}.forward needs the noise level t")
            film = self._film(t_dev, ub, N)
            # Tensor formats of the default path (every MFMA layer on the split-operand kernel, fp32 results):
            #   x   (block input: conv_in / stride-2 / decoder GEMM output; read by conv1 and as conv2's residual)  planes of 4 channels
            #   tmp (conv1 -> conv2)      split planes of SiLU(FiLM(conv1)), stored by conv1's epilogue
            #   out (block output; read by the stride-2 layer and, as `cur` / skip, by the decoder GEMMs)  split planes, raw
            # so conv2, the stride-2 layers and the decoder GEMMs stage by LDS-DMA alone and every epilogue but the last stores
            # from the accumulator layout.  The last block keeps [N][H][W][C]: its conv2 carries the fused output projection.
            flow = self._sp_flow(N, H, W)
            pt = 1 if (flow and getattr(self, 'precision', 'fp32') == 'fp16') else 2     # parts of the split-plane tensors: 1 = h-only planes (fp16 path)
            P4, SP = 2, 1
            a = self._new(N, H, W, nfp)
            L.check(lib.yond_conv_in_f32(L.ptr(x4), L.ptr(ub), N, H, W, nfp, L.ptr(self.conv_in_w), L.ptr(self.conv_in_b),
                                         0.01, L.ptr(a), P4 if flow else 0, st), "yond_conv_in_f32")
            skips = {}
            h, w = H, W
            cur = a
            xsp = None                                           # SiLU(x) in split planes, stored by the stride-2 layer that produced x
            for i in range(1, 10):
                blk = self.blocks[i]
                cp = blk['Cp']
                f = film[f'conv{i}']
                last = i == 9
                xfmt = P4 if (flow and not last) else 0          # format of this block's input x
                if i >= 6:
                    # ConvT 2x2 s2 + the block's 1x1 shortcut over [up, skip] as one GEMM with a pixel-shuffle store
                    xs = self._new(N, 2 * h, 2 * w, cp)
                    up = blk['upsc2'] if (flow and K1_SUB2 and 'upsc2' in blk) else blk['upsc']
                    # (decoder level of block i = 9 - i; images too narrow for the unfolded GEMM keep the folded kernels, which have no second output)
                    xsp = (self._new_sp(('xspd', i), N, 2 * h, 2 * w, cp, pt)
                           if (flow and not last and (9 - i) in (K1_D2_LEVELS if pt == 2 else K1_D2_LEVELS_HALF) and up is blk['upsc'] and cp % 64 == 0 and w > 16) else None)     # (the dispatcher's second output: 64-wide tiles, conv_split.hip)
                    self._conv(up, cur, skips[10 - i], N, h, w, xs, in_fmt=SP if flow else 0, out_fmt=xfmt, dst2=xsp)
                    h, w = 2 * h, 2 * w
                    cur = xs
                # z = conv2(SiLU(FiLM(conv1(SiLU(x))))) + x : the first SiLU runs in conv1's staging (x has other readers); the
                # second in conv1's EPILOGUE where both layers run on the split kernel -- tmp has one reader, whose staging
                # would repeat it once per output-channel tile (bit-identical: the same fp32 function of the same value).
                # At split precision tmp is stored in SPLIT PLANES: conv1 writes the (h, l) halves conv2 would have staged,
                # conv2 stages them by LDS-DMA alone (the same bits again)
                if flow and FUSE_BLOCK0 and h == H and cp == 32 and self._block0_weights(blk['conv1']) is not None and self._block0_weights(blk['conv2']) is not None:
                    # level 0: the whole block in one launch -- tmp never leaves the chip (csrc/block0_fused.hip)
                    if last and self._out4_fusable(blk['conv2'], flow):
                        out4 = self._new(N, H, W, 4)
                        self._block0(blk['conv1'], blk['conv2'], cur, N, h, w, f, xfmt, out4=(self.w_out, self.b_out, x4 if self.res else None, ub, out4))
                        return out4
                    if not last:
                        out = self._new_sp(('out', i), N, h, w, cp, pt)
                        self._block0(blk['conv1'], blk['conv2'], cur, N, h, w, f, xfmt, dst=out)
                        cur = out
                        skips[i] = cur
                        nxt = self._new(N, h // 2, w // 2, blk['pool'].coutp)
                        xsp = self._new_sp(('xsp', i + 1), N, h // 2, w // 2, blk['pool'].coutp, pt) if (flow and i >= SP_CONV1_MIN_LEVEL) else None
                        self._conv(blk['pool'], cur, None, N, h, w, nxt, in_fmt=SP, out_fmt=P4, dst2=xsp)
                        h, w = h // 2, w // 2
                        cur = nxt
                        continue
                act_in_producer = self._split_pair(blk['conv1'], blk['conv2'])
                sp = SP if (flow or (act_in_producer and SPLIT_PLANES and getattr(self, 'precision', 'fp32') == 'fp32'
                                      and sp_plane_units(h, w) * 64 < 2 ** 31)) else 0
                tmp = self._new_sp(('tmp', i), N, h, w, cp, pt) if sp else self._new(N, h, w, cp)
                if xsp is not None and sp and act_in_producer:
                    # conv1's input as the producer of x stored it: SiLU applied, split -- staged by LDS-DMA alone
                    self._conv(blk['conv1'], xsp, None, N, h, w, tmp, escale=f[0], eshift=f[1], ebatch=1, pre_act=0, post_act=1,
                               in_fmt=SP, out_fmt=sp)
                else:
                    self._conv(blk['conv1'], cur, None, N, h, w, tmp, escale=f[0], eshift=f[1], ebatch=1, pre_act=1,
                               post_act=1 if act_in_producer else 0, in_fmt=xfmt, out_fmt=sp)
                xsp = None
                pre2 = 0 if act_in_producer else 1
                if last and self._out4_fusable(blk['conv2'], flow):
                    # the last block's output feeds only the 1x1 output projection: computed in this epilogue, never stored
                    out4 = self._new(N, H, W, 4)
                    self._conv(blk['conv2'], tmp, None, N, h, w, None, escale=f[2], eshift=f[3], ebatch=1, res=cur, pre_act=pre2,
                               out4=(self.w_out, self.b_out, x4 if self.res else None, ub, out4), in_fmt=sp)
                    return out4
                if flow:
                    out = self._new_sp(('out', i), N, h, w, cp, pt)
                    self._conv(blk['conv2'], tmp, None, N, h, w, out, escale=f[2], eshift=f[3], ebatch=1, res=cur, pre_act=pre2,
                               in_fmt=SP, out_fmt=SP, res_fmt=P4)
                else:
                    out = self._new(N, h, w, cp)
                    self._conv(blk['conv2'], tmp, None, N, h, w, out, escale=f[2], eshift=f[3], ebatch=1, res=cur, pre_act=pre2, in_fmt=sp)
                cur = out
                if i <= 4:
                    skips[i] = cur
                    nxt = self._new(N, h // 2, w // 2, blk['pool'].coutp)
                    xsp = self._new_sp(('xsp', i + 1), N, h // 2, w // 2, blk['pool'].coutp, pt) if (flow and i >= SP_CONV1_MIN_LEVEL) else None   # (level of block i + 1 = i)
                    self._conv(blk['pool'], cur, None, N, h, w, nxt, in_fmt=SP if flow else 0, out_fmt=P4 if flow else 0, dst2=xsp)
                    h, w = h // 2, w // 2
                    cur = nxt
            feat = cur
        else:
            a = self._new(N, H, W, nfp)
            L.check(lib.yond_conv_in_f32(L.ptr(x4), L.ptr(ub), N, H, W, nfp, L.ptr(self.conv_in_w), L.ptr(self.conv_in_b),
                                         0.2, L.ptr(a), 0, st), "yond_conv_in_f32")
            cv = self.convs
            h, w = H, W
            cur = self._conv(cv['conv1_2'], a, None, N, h, w, self._new(N, h, w, cv['conv1_2'].coutp), post_act=2, slope=0.2)
            skips = {1: cur}

            def first_of_pair(i, c1, c2, s0, s1):
                """conv{i}_1 -> LeakyReLU: its one reader is conv{i}_2, so at split precision the tensor travels in split planes
                (the producer applies the activation and the split, the consumer stages by LDS-DMA alone: the same bits).
                Returns (tensor, in_fmt for conv{i}_2)."""
                sp = (UNET_SP and SPLIT_PLANES and not getattr(self, 'strict', False) and getattr(self, 'conv_algo', WINO_DEFAULT) == 'split'
                      and getattr(self, 'precision', 'fp32') == 'fp32' and c1.split(2) is not None and c2.split(2) is not None
                      and N * sp_plane_units(h, w) * 64 < 2 ** 31)
                dst = self._new_sp(('utmp', i), N, h, w, c1.coutp) if sp else self._new(N, h, w, c1.coutp)
                self._conv(c1, s0, s1, N, h, w, dst, post_act=2, slope=0.2, out_fmt=1 if sp else 0)
                return dst, (1 if sp else 0)

            for i in range(2, 6):
                cpv = cur.shape[-1]
                pooled = self._new(N, h // 2, w // 2, cpv)
                L.check(lib.yond_maxpool2_f32(L.ptr(cur), N, h, w, cpv, L.ptr(pooled), st), "yond_maxpool2_f32")
                h, w = h // 2, w // 2
                c1, c2 = cv[f'conv{i}_1'], cv[f'conv{i}_2']
                cur, fmt = first_of_pair(i, c1, c2, pooled, None)
                cur = self._conv(c2, cur, None, N, h, w, self._new(N, h, w, c2.coutp), post_act=2, slope=0.2, in_fmt=fmt)
                if i < 5:
                    skips[i] = cur
            for i in range(6, 10):
                upc = cv[f'upv{i}']
                up = self._new(N, 2 * h, 2 * w, upc.cout_real_p)
                self._conv(upc, cur, None, N, h, w, up)
                h, w = 2 * h, 2 * w
                c1, c2 = cv[f'conv{i}_1'], cv[f'conv{i}_2']
                cur, fmt = first_of_pair(i, c1, c2, up, skips[10 - i])
                if i == 9 and self._out4_fusable(c2):
                    out4 = self._new(N, H, W, 4)
                    self._conv(c2, cur, None, N, h, w, None, post_act=2, slope=0.2, in_fmt=fmt,
                               out4=(self.w_out, self.b_out, x4 if self.res else None, ub, out4))
                    return out4
                cur = self._conv(c2, cur, None, N, h, w, self._new(N, h, w, c2.coutp), post_act=2, slope=0.2, in_fmt=fmt)
            feat = cur
        out4 = self._new(N, H, W, 4)
        L.check(lib.yond_conv_out_f32(L.ptr(feat), feat.shape[-1], L.ptr(self.w_out), L.ptr(self.b_out),
                                      L.ptr(x4) if self.res else None, L.ptr(ub), N, H, W, L.ptr(out4), st), "yond_conv_out_f32")
        return out4

    def forward_nchw(self, x, t=None):
        """Plugin-surface call: x [N][4][H][W] -> [N][4][H][W]."""
        L.require_cuda(x, "x")
        N, c, H, W = x.shape
        if c != 4:
            raise L.YondHipError(f"expected 4 input channels, got {c}")
        x4 = torch.empty((N, H, W, 4), dtype=torch.float32, device=x.device)
        L.check(self.lib.yond_nchw4_to_nhwc4_f32(L.ptr(x), L.ptr(x4), N, H, W, L.stream()), "yond_nchw4_to_nhwc4_f32")
        t_dev = None
        if self.guided:
            t_dev = torch.as_tensor(t, dtype=torch.float32, device=x.device).reshape(-1)
            if t_dev.numel() == 1 and N > 1:
                t_dev = t_dev.expand(N)
            if t_dev.numel() != N:
                raise L.YondHipError(f"t must have 1 or {N} elements, got {t_dev.numel()}")
            t_dev = t_dev.contiguous()
        # Plugin surface = the place a caller cannot be asked to bracket the call: the range guard of the half-precision
        # operand paths runs here (status slot 3; the pipeline wrappers use slots 0-2): the forward, one read of the status
        # word (this surface is synchronous for its callers anyway -- the reference's `.cpu()` follows), and if a staged
        # activation left fp16's range the forward is recomputed on the fp32-input MFMA kernels.
        guarded = self.uses_half_operands()
        if guarded:
            self.begin_guard(3)
        y4 = self.forward_nhwc4(x4, t_dev)
        if guarded and self.overflowed(3):
            import warnings
            warnings.warn("an activation left fp16's range (|a| > 65504) in the split-operand convolution path: "
                          "this forward is recomputed on the fp32-input MFMA kernels")
            self.strict = True
            try:
                y4 = self.forward_nhwc4(x4, t_dev)
            finally:
                self.strict = False
        y = torch.empty_like(x)
        L.check(self.lib.yond_nhwc4_to_nchw4_f32(L.ptr(y4), L.ptr(y), N, H, W, L.stream()), "yond_nhwc4_to_nchw4_f32")
        return y
