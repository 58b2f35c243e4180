"""N4: the training step of the AWGN raw denoiser on the MI355X kernels, its learning-rate schedule, the epoch loop and the
data-parallel gradient exchange.

    trainer_AWGN.py:101-117      pred = net(imgs_lr, sigma); loss = Unet_Loss()(pred, imgs_hr); loss.backward(); optimizer.step()
    losses/base_loss.py:69-113   Unet_Loss = F.l1_loss, or L1_Charbonnier_loss with charbonnier=True
    trainer_AWGN.py:36           Adam(net.parameters(), lr)
    trainer_base.py:34-46, 138-167   LambdaScheduler over get_cos_lr / get_multistep_lr, stepped once per epoch (trainer_AWGN.py:56-57, 153)
    trainer_AWGN.py:59-61        DistributedDataParallel: gradients averaged over the ranks (distributed.GradReducer, RCCL over xGMI)

`TrainStep` runs GuidedResUnet's forward (archs/Unet.py:424-470) on NHWC float32 device tensors whose channels are padded
to multiples of 32, with every convolution -- forward, data gradient and weight gradient -- on the HIP kernels:
  * forward / data gradient: yond_conv2d_f32, algo 0 (fp32-input MFMA, bit-exact fmaf chains).  A data gradient is a
    convolution with re-indexed weights: 3x3 stride 1 -> 3x3 with the taps flipped and (in, out) transposed; 3x3 stride 2 ->
    the same on the zero-interleaved gradient; ConvTranspose2d 2x2 -> a 1x1 GEMM on the pixel-unshuffled gradient; 1x1 -> W^T;
  * weight / bias gradient: yond_conv_wgrad_f32 (fp32 MFMA outer products over the pixel axis), yond_colsum_f32;
  * loss and optimiser: yond_l1_loss_f32, yond_adam_step_f32.
torch.autograd only strings the layers together (custom Functions) and differentiates the elementwise glue (SiLU, LeakyReLU,
FiLM scale / shift and its three tiny sigma-MLPs, residual adds), which is <0.1 % of the step's arithmetic.
Scope: GuidedResUnet and UNetSeeInDark (the two nets the reference's runfiles/Gaussian/*.yml train) / fp32; weights are re-packed on the host every step (fine at nf = 8 ... 32; a production loop would keep
packed weights resident).  Data-parallel: one process per GPU, every rank steps on its own batches, `GradReducer` averages the
gradients in two 25 MB buckets launched from backward's hooks (the only collective of training: 44.7 MB per step).
Pinned by tests/golden/train.npz (loss, gradients and updated weights of the reference's own step on the same inputs) and
tests/golden/train_sched.npz (the reference's schedules, a two-epoch run of its loop, its Charbonnier loss)."""
import ctypes as C
import math
import types

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L
from .engine import DenoiserPlan, _PackedConv, _rup


WGRAD_BIAS = True                # the bias gradient of a 3x3 stride-1 layer rides along in the split-operand weight-gradient kernel
FILM_ALL = True                     # the guided blocks' sigma-MLPs in one forward and one backward call for the whole net (False: per block)
FLAT_1X1 = True                  # 1x1 convolutions over images narrower than 32 pixels run on a [1][P/32][32][C] view of the same memory
GEMM_SPLIT = True                # 1x1 / transposed convolutions and their data gradients as split-operand GEMMs on the fp16 matrix cores (False: fp32 MFMA)
GEMM_MIN_K = 256                 # ... from this many input channels on (the compute-bound levels)
                                 # (module attribute: tools/ flip it for A/B runs)


def _plan(dev):
    plan = DenoiserPlan.__new__(DenoiserPlan)
    plan.lib, plan.dev = L.load(), torch.device(dev)
    plan.arena = None                # TrainStep: every parameter in one flat device buffer, element 0 a constant zero
    plan.wcache = {}                 # (parameter, role, shape) -> packed-layout constants (_Packing), built once
    # range guard of the split-operand kernels (bit 0: a staged activation / gradient, bit 1 here: a weight left fp16's range):
    # word 0 is what the convolution launches and the weight packer report into; TrainStep reads it once per step
    plan.status = torch.zeros(4, dtype=torch.int32, device=plan.dev)
    plan.status_slot = 0
    plan.gen = 0                     # bumped whenever a buffer the step's launches point at is replaced: captured steps are then stale
    plan.wgrad_stream = torch.cuda.Stream(device=plan.dev) if plan.dev.type == 'cuda' else None      # the weight gradients' lane (_lane)
    return plan


def _pad_c(x, cp):
    """[N][H][W][C] -> channels zero-padded to cp."""
    return x if x.shape[-1] == cp else F.pad(x, (0, cp - x.shape[-1]))


class _Packing:
    """The device-side constants of one convolution whose weights live in the parameter arena: the kernel's packed weight layout
    is a fixed gather of arena elements (index 0 = the arena's zero, for the channel padding), computed ONCE by pushing the
    parameter's element numbers through the same re-indexing (`xf`) and the same host packer the weights themselves would take
    (element numbers stay exact in float32: the largest layer has 2.4 M weights < 2^24).  Every step re-packs with one gather on
    the device; nothing crosses the host."""

    def __init__(self, plan, w, b, xf, ksize, stride, splits, shuffle, N, Ho, Wo):
        arena = plan.arena
        base = (w.data_ptr() - arena.data_ptr()) // 4
        assert 0 < base and base + w.numel() <= arena.numel() and w.numel() < (1 << 24)
        ids = torch.arange(1, w.numel() + 1, dtype=torch.float32).reshape(w.shape)
        self.pc = _PackedConv(plan.dev, xf(ids), None, ksize, stride, splits, shuffle=shuffle)
        tn, kc, packed_ids = self.pc.config(N, Ho, Wo)
        ids = packed_ids.round().long()
        self.key = (tn, kc, False)
        self.wmap = torch.where(ids > 0, ids - 1 + base, torch.zeros_like(ids))
        # the split-operand kernel (fp16 matrix cores, fp32-accurate products) where it takes the layer: its packing holds (h, l)
        # half pairs, not a permutation -- the padded OIHW weights are gathered, then packed by yond_pack_conv_split_weight_dev_f32
        self.split_tn = 0
        if getattr(plan, 'train_conv', 'split') == 'split' and bool(shuffle) == (ksize == 1):
            self.split_tn = int(plan.lib.yond_conv_split_supported(ksize, stride, self.pc.cinp, self.pc.gemm_n))
        if self.split_tn:
            oid = torch.from_numpy(self.pc._wp.reshape(-1)).round().long()
            self.omap = torch.where(oid > 0, oid - 1 + base, torch.zeros_like(oid)).to(plan.dev)
        self.status = plan.status[1:2] if getattr(plan, 'status', None) is not None else None    # (word 1: weights out of range)
        self.bmap = None
        if b is not None:
            bb = (b.data_ptr() - arena.data_ptr()) // 4
            bm = torch.zeros(self.pc.coutp, dtype=torch.long)
            bm[:b.numel()] = torch.arange(bb, bb + b.numel())
            self.bmap = bm.to(plan.dev)

    def maps(self):
        """The gathers of this layer: the weight map and, when the layer has one, the bias map (TrainStep batches them)."""
        return [self.omap if self.split_tn else self.wmap] + ([self.bmap] if self.bmap is not None else [])

    def refresh(self, arena, lib, batch=None):
        """batch: (buffer, {id(map): (offset, length)}) -- the step's ONE gather of every layer's maps; None: gather here."""
        def gathered(m):
            if batch is not None and id(m) in batch[1]:
                o, n = batch[1][id(m)]
                return batch[0][o:o + n]
            return arena.index_select(0, m)
        if self.split_tn and batch is not None and len(batch) > 2 and id(self) in batch[2]:
            pass                                             # packed by the step's one batched launch (TrainStep._gather_weights)
        elif self.split_tn:
            wp = gathered(self.omap)
            # pack INTO the tensor the layer already reads when there is one: it may be a slice of the step's batched destination buffer
            # (TrainStep._gather_weights re-points the slices only when the set of layers changes), and a private replacement would leave
            # the kernels on stale weights from the next batched pack on
            cur = self.pc._packed.get(('split', 2))
            packed = cur[1] if cur is not None and cur[0] == self.split_tn and cur[1].numel() == wp.numel() else torch.empty_like(wp)
            st = getattr(self, 'status', None)
            L.check(lib.yond_pack_conv_split_weight_dev_f32(L.ptr(wp), self.pc.gemm_n, self.pc.cinp, self.pc.ksize, self.split_tn, 2,
                                                            L.ptr(packed), L.ptr(st), L.stream()), "yond_pack_conv_split_weight_dev_f32")
            self.pc._packed[('split', 2)] = (self.split_tn, packed)
        else:
            self.pc._packed[self.key] = gathered(self.wmap)
        if self.bmap is not None:
            self.pc.bias = gathered(self.bmap)
        return self.pc


def _conv_fwd(plan, w, b, ksize, stride, splits, srcs, N, H, W, shuffle=False, role='fwd', xf=None, res=None):
    """One convolution launch with `xf(w)` as its OIHW weights (xf: the re-indexing that turns the parameter into this launch's
    weights -- identity for a forward, flip + transpose for a data gradient; it must work on CPU and device tensors alike)."""
    xf = xf if xf is not None else (lambda t: t)
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)
    if plan.arena is not None:
        key = (w.data_ptr(), role, ksize, stride, tuple(splits), bool(shuffle), N, H, W)
        pk = plan.wcache.get(key)
        if pk is None:
            pk = plan.wcache[key] = _Packing(plan, w, b, xf, ksize, stride, splits, shuffle, N, Ho, Wo)
        pc = pk.refresh(plan.arena, plan.lib, getattr(plan, 'wbatch', None))
        algo = 'split' if pk.split_tn else 0
    else:                            # a bare plan (kernel tests): pack on the host
        pc = _PackedConv(plan.dev, xf(w.detach()).cpu(), None if b is None else b.detach().cpu(), ksize, stride, splits, shuffle=shuffle)
        algo = 0
    if shuffle:
        out = torch.empty((N, 2 * H, 2 * W, pc.cout_real_p), dtype=torch.float32, device=plan.dev)
    else:
        out = torch.empty((N, Ho, Wo, pc.coutp), dtype=torch.float32, device=plan.dev)
    plan._conv(pc, srcs[0], srcs[1] if len(srcs) > 1 else None, N, H, W, out, algo=algo, res=res)
    return out


def _wgrad(plan, x, dy, mode, stride, taps, with_bias=False, oihw=False):
    """[taps][Cout_p][Cin_p] float32 (x, dy: padded NHWC).  The workgroups' partial sums go through one workspace that grows to the
    largest layer's need (stream order keeps its uses apart).  with_bias: returns (dw, db [Cout_p]) -- the split-operand kernel adds
    up dy's columns on the way, the other layers take yond_colsum_f32."""
    if with_bias:
        N, H, W, ci = x.shape
        co = dy.shape[-1]
        if mode == 0 and stride == 1 and getattr(plan, 'train_conv', 'split') == 'split':
            need = int(plan.lib.yond_conv_wgrad_split_ws_bytes(N, H, W, ci, co))
            if need:
                ws = getattr(plan, 'wgrad_ws', None)
                if ws is None or ws.numel() * 4 < need:
                    ws = plan.wgrad_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device); plan.gen += 1   # (captured steps hold the old one)
                buf = torch.empty(taps * co * ci + co, dtype=torch.float32, device=x.device)
                L.check(plan.lib.yond_conv_wgrad_split_f32(L.ptr(x), L.ptr(dy), N, H, W, ci, co, L.ptr(buf), 3 if oihw else 1, L.ptr(ws),
                                                           ws.numel() * 4, L.ptr(getattr(plan, 'status', None)), L.stream()),
                        "yond_conv_wgrad_split_f32")
                if oihw:                                     # [co][ci][3][3], the parameter's own order: no permute / copy pass
                    return buf[:taps * co * ci].view(co, ci, 3, 3), buf[taps * co * ci:]
                return buf[:taps * co * ci].view(taps, co, ci), buf[taps * co * ci:]
        dw = _wgrad(plan, x, dy, mode, stride, taps)
        return (dw.permute(1, 2, 0).reshape(co, ci, 3, 3) if oihw else dw), _colsum(plan, dy)
    N, H, W, ci = x.shape
    _, Ho, Wo, co = dy.shape
    dw = torch.empty((taps, co, ci), dtype=torch.float32, device=x.device)
    if mode == 0 and stride == 1 and getattr(plan, 'train_conv', 'split') == 'split':
        # 3x3 stride 1 (18 of a step's 27 weight gradients, 95 % of their FLOPs): fp32-accurate split operands on the fp16 MFMA
        need = int(plan.lib.yond_conv_wgrad_split_ws_bytes(N, H, W, ci, co))
        if need:
            ws = getattr(plan, 'wgrad_ws', None)
            if ws is None or ws.numel() * 4 < need:
                ws = plan.wgrad_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device); plan.gen += 1   # (captured steps hold the old one)
            st = getattr(plan, 'status', None)
            L.check(plan.lib.yond_conv_wgrad_split_f32(L.ptr(x), L.ptr(dy), N, H, W, ci, co, L.ptr(dw), 0, L.ptr(ws), ws.numel() * 4,
                                                       L.ptr(st), L.stream()), "yond_conv_wgrad_split_f32")
            return dw
    need = int(plan.lib.yond_conv_wgrad_ws_bytes(N, H, W, ci, Ho, Wo, co, mode, stride))
    ws = getattr(plan, 'wgrad_ws', None)
    if ws is None or ws.numel() * 4 < need:
        ws = plan.wgrad_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device); plan.gen += 1   # (captured steps hold the old one)
    L.check(plan.lib.yond_conv_wgrad_ws_f32(L.ptr(x), L.ptr(dy), N, H, W, ci, Ho, Wo, co, mode, stride, L.ptr(dw), L.ptr(ws), ws.numel() * 4,
                                            L.stream()), "yond_conv_wgrad_ws_f32")
    return dw


WGRAD_LANE = True                    # a layer's weight (and bias) gradient on a second HIP stream beside its data gradient: fork / join inside every backward,
                                     # so the captured step has two branches per layer -- one kernel's launch gap and partly filled last round under the other's work


class _lane:
    """`with _lane(plan):` -- the launches inside go to the plan's second stream, ordered behind everything queued on the current one so far (fork);
    `_join(plan)` orders the current stream behind them.  Every backward that forks joins before it returns: autograd sees finished tensors, the
    allocator's per-stream pools stay safe (a block of the lane's pool is only handed out again on the lane, behind the next fork), and a stream
    capture ends with the lane joined."""

    def __init__(self, plan):
        self.on = WGRAD_LANE and plan.dev.type == 'cuda' and getattr(plan, 'wgrad_stream', None) is not None
        self.plan = plan

    def __enter__(self):
        if self.on:
            side = self.plan.wgrad_stream
            side.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
            self.plan.lane_open = True
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False


def _join(plan):
    if getattr(plan, 'lane_open', False):
        torch.cuda.current_stream().wait_stream(plan.wgrad_stream)
        plan.lane_open = False


def _colsum(plan, dy):
    c = dy.shape[-1]
    db = torch.empty(c, dtype=torch.float32, device=dy.device)
    L.check(plan.lib.yond_colsum_f32(L.ptr(dy), dy.numel() // c, c, L.ptr(db), L.stream()), "yond_colsum_f32")
    return db


class _Conv3x3(torch.autograd.Function):
    """nn.Conv2d(cin, cout, 3, stride, 1): x [N][H][W][cin_p] -> [N][Ho][Wo][cout_p]."""

    @staticmethod
    def forward(ctx, x, w, b, plan, stride, need_dx, res=None):
        """res: a tensor of the output's shape added in the convolution's epilogue (the residual of a guided block: `z += x`,
        archs/modules.py:195) -- its gradient is dy itself."""
        N, H, W, _ = x.shape
        cout, cin = w.shape[0], w.shape[1]
        x = x.contiguous()
        y = _conv_fwd(plan, w, b, 3, stride, [cin], [x], N, H, W, res=None if res is None else res.contiguous())
        ctx.save_for_backward(x, w)
        ctx.plan, ctx.stride, ctx.need_dx, ctx.has_res = plan, stride, need_dx, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        plan, stride = ctx.plan, ctx.stride
        dy = dy.contiguous()
        N, H, W, cin_p = x.shape
        cout, cin = w.shape[0], w.shape[1]
        with _lane(plan):
            if WGRAD_BIAS:
                dw, db = _wgrad(plan, x, dy, 0, stride, 9, with_bias=True, oihw=True)
                dw = dw if (dw.shape[0] == cout and dw.shape[1] == cin) else dw[:cout, :cin].contiguous()
            else:
                dw, db = _wgrad(plan, x, dy, 0, stride, 9), _colsum(plan, dy)
                dw = dw[:, :cout, :cin].permute(1, 2, 0).reshape(cout, cin, 3, 3)
            db = db[:cout]
        dx = None
        if ctx.need_dx:
            g = dy
            if stride == 2:                            # zero-interleave: the stride-2 layer's adjoint = a stride-1 one on this
                g = torch.empty((N, H, W, dy.shape[-1]), dtype=torch.float32, device=dy.device)
                L.check(plan.lib.yond_zero_interleave_f32(L.ptr(dy), N, dy.shape[1], dy.shape[2], dy.shape[-1], H, W, L.ptr(g), L.stream()),
                        "yond_zero_interleave_f32")
            dx = _conv_fwd(plan, w, None, 3, 1, [cout], [g], N, H, W, role='dgrad',
                           xf=lambda t: t.flip(2, 3).transpose(0, 1).contiguous())          # [cin][cout][2-ky][2-kx]
            dx = _pad_c(dx[..., :cin], cin_p) if dx.shape[-1] != cin_p else dx
        _join(plan)
        return dx, dw, db, None, None, None, (dy if ctx.has_res else None)


class _Conv3x3Cat(torch.autograd.Function):
    """nn.Conv2d(c0 + c1, cout, 3, 1, 1) over torch.cat([x0, x1], channel) (UNetSeeInDark's conv{6..9}_1, archs/Unet.py:83-99):
    the concatenation is never materialised -- forward as a two-source convolution, the weight gradient per source, each data
    gradient a convolution of dy with that source's slice of the flipped, transposed weights."""

    @staticmethod
    def forward(ctx, x0, x1, w, b, plan, c0):
        N, H, W, _ = x0.shape
        x0, x1 = x0.contiguous(), x1.contiguous()
        c1 = w.shape[1] - c0
        y = _conv_fwd(plan, w, b, 3, 1, [c0, c1], [x0, x1], N, H, W)
        ctx.save_for_backward(x0, x1, w)
        ctx.plan, ctx.c0 = plan, c0
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, x1, w = ctx.saved_tensors
        plan, c0 = ctx.plan, ctx.c0
        dy = dy.contiguous()
        N, H, W, _ = dy.shape
        cout, c1 = w.shape[0], w.shape[1] - c0
        dws, dxs = [], []
        with _lane(plan):
            for x, lo, c in ((x0, 0, c0), (x1, c0, c1)):
                dws.append(_wgrad(plan, x, dy, 0, 1, 9)[:, :cout, :c].permute(1, 2, 0).reshape(cout, c, 3, 3))
            db = _colsum(plan, dy)[:cout]
            dw = torch.cat(dws, 1)
        for x, lo, c in ((x0, 0, c0), (x1, c0, c1)):
            dx = _conv_fwd(plan, w, None, 3, 1, [cout], [dy], N, H, W, role=('dgrad', lo),
                           xf=lambda t, lo=lo, c=c: t[:, lo:lo + c].flip(2, 3).transpose(0, 1).contiguous())    # [c][cout][2-ky][2-kx]
            dxs.append(dx if dx.shape[-1] == x.shape[-1] else _pad_c(dx[..., :c], x.shape[-1]))
        _join(plan)
        return dxs[0], dxs[1], dw, db, None, None


def _gemm_split(plan, srcs, P, n_p, n_real, sn_lo, sn_hi, nblk, bias, y, ldy, shuffle=0, H=0, W=0):
    """yond_gemm_split_f32: y[p][n] = sum_s sum_k x_s[p][k] B_s(k, n) (+ bias) with the float32 parameter read in place through strides
    (include/yond_hip.h).  srcs: (x, w_ptr, sk_lo, sk_hi, ld, k, kblk, k_real) per source; a weight beyond +-32 sets the plan's weight word."""
    arr = (L.GemmSrc * len(srcs))()
    for d, (x, wp, sk_lo, sk_hi, ld, k, kblk, k_real) in zip(arr, srcs):
        d.x, d.w, d.sk_lo, d.sk_hi, d.ld, d.k, d.kblk, d.k_real = x.data_ptr(), wp, sk_lo, sk_hi, ld, k, kblk, k_real
    L.check(plan.lib.yond_gemm_split_f32(C.cast(arr, C.c_void_p), len(srcs), P, n_p, n_real, sn_lo, sn_hi, nblk, None if bias is None else L.ptr(bias),
                                         L.ptr(y), ldy, shuffle, H, W, L.ptr(plan.status[0:2]), L.stream()), "yond_gemm_split_f32")
    return y


def _use_gemm(plan, K):
    """The split-operand GEMM where it measures faster than the fp32-MFMA convolution path (tools/gemm_ab.py, device times at the training
    shape): the deep, compute-bound levels -- at least GEMM_MIN_K input channels in all (a 1x1 over 2 x 256: 42 vs 69 us, the transposed
    512 -> 256: 43 vs 184 us); the wide, shallow levels (K <= 128: 2-4 K steps per tile) stay where they are (74 vs 66, 132 vs 112 us)."""
    return (GEMM_SPLIT and K >= GEMM_MIN_K and plan.arena is not None and getattr(plan, 'train_conv', 'split') == 'split'
            and not getattr(plan, 'gemm_off', False))      # (gemm_off: a weight with |w| >= 32 met the GEMM's 2^11 w part -- see _weight_trip)


_BIG = 1 << 30


def _flat32(t):
    """A 1x1 convolution has no spatial structure: images narrower than the kernels' 32-pixel tile rows (the deep levels of a batch of
    small patches: 16 and 8 pixels wide) are handed over as ONE image of 32-pixel rows -- a view of the same contiguous memory -- so
    that no tile is half or three quarters padding."""
    N, H, W, C = t.shape
    P = N * H * W
    return t.view(1, P // 32, 32, C) if (FLAT_1X1 and W < 32 and P % 32 == 0 and t.is_contiguous()) else t


class _Conv1x1(torch.autograd.Function):
    """nn.Conv2d(c0 + c1, cout, 1) over the channel concatenation of one or two tensors (torch.cat is not materialised)."""

    @staticmethod
    def forward(ctx, x0, x1, w, b, plan):
        shape = x0.shape
        splits = [w.shape[1]] if x1 is None else list(ctx_splits(w, x0, x1))
        if _use_gemm(plan, sum(t.shape[-1] for t in (x0, x1) if t is not None)):   # a plain GEMM over the pixels: split operands on the fp16 matrix cores
            srcs = [x0.contiguous()] + ([x1.contiguous()] if x1 is not None else [])
            cout, P = w.shape[0], shape[0] * shape[1] * shape[2]
            n_p, off, gs = _rup(cout), 0, []
            for x, c in zip(srcs, splits):
                gs.append((x, w.data_ptr() + 4 * off, 1, 0, x.shape[-1], x.shape[-1], _BIG, c))
                off += c
            y = torch.empty((shape[0], shape[1], shape[2], n_p), dtype=torch.float32, device=x0.device)
            _gemm_split(plan, gs, P, n_p, cout, w.shape[1], 0, _BIG, b, y, n_p)
            ctx.save_for_backward(*srcs, w)
            ctx.plan, ctx.splits, ctx.two, ctx.shape = plan, splits, x1 is not None, shape
            return y
        srcs = [_flat32(x0.contiguous())] + ([_flat32(x1.contiguous())] if x1 is not None else [])
        N, H, W, _ = srcs[0].shape
        y = _conv_fwd(plan, w, b, 1, 1, splits, srcs, N, H, W)
        ctx.save_for_backward(*srcs, w)
        ctx.plan, ctx.splits, ctx.two, ctx.shape = plan, splits, x1 is not None, shape
        return y.view(shape[0], shape[1], shape[2], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        *srcs, w = ctx.saved_tensors
        plan, splits, shape = ctx.plan, ctx.splits, ctx.shape
        if _use_gemm(plan, dy.shape[-1]):
            dy = dy.contiguous()
            cout, P, n_p = w.shape[0], shape[0] * shape[1] * shape[2], dy.shape[-1]
            dws, dxs, off = [], [], 0
            srcs = [x.view(shape[0], shape[1], shape[2], x.shape[-1]) for x in srcs]    # (the forward may have run on the flat view)
            xf, dyf = [_flat32(x) for x in srcs], _flat32(dy)                       # (the weight gradient's rows: 32 pixels wide)
            with _lane(plan):
                for x, xw, c in zip(srcs, xf, splits):
                    dws.append(_wgrad(plan, xw, dyf, 2, 1, 1)[0, :cout, :c])
                dw = torch.cat(dws, 1)[:, :, None, None]
                db = _colsum(plan, dy)[:cout]
            for x, xw, c in zip(srcs, xf, splits):
                dx = torch.empty_like(x)                                            # dx[p][ci] = sum_co dy[p][co] w[co][off + ci]
                _gemm_split(plan, [(dy, w.data_ptr() + 4 * off, w.shape[1], 0, n_p, n_p, _BIG, cout)], P, x.shape[-1], c, 1, 0, _BIG, None,
                            dx, x.shape[-1])
                dxs.append(dx)
                off += c
            _join(plan)
            return dxs[0], (dxs[1] if ctx.two else None), dw, db, None
        dy = _flat32(dy.contiguous()) if srcs[0].shape[:3] != shape[:3] else dy.contiguous()      # (the geometry the forward ran in)
        N, H, W, _ = dy.shape
        cout = w.shape[0]
        dws, dxs, off = [], [], 0
        with _lane(plan):
            for x, s in zip(srcs, splits):
                dws.append(_wgrad(plan, x, dy, 2, 1, 1)[0, :cout, :s])
            dw = torch.cat(dws, 1)[:, :, None, None]
            db = _colsum(plan, dy)[:cout]
        for x, s in zip(srcs, splits):
            dx = _conv_fwd(plan, w, None, 1, 1, [cout], [dy], N, H, W, role=('dgrad', off),
                           xf=lambda t, off=off, s=s: t[:, off:off + s, 0, 0].t().contiguous()[:, :, None, None])   # [s][cout][1][1]
            dx = dx if dx.shape[-1] == x.shape[-1] else _pad_c(dx[..., :s], x.shape[-1])
            dxs.append(dx.view(shape[0], shape[1], shape[2], dx.shape[-1]))
            off += s
        _join(plan)
        return dxs[0], (dxs[1] if ctx.two else None), dw, db, None


class _FilmSilu(torch.autograd.Function):
    """SiLU(z * tk + tb) with per-image (scale, shift) vectors (archs/modules.py:186-196), forward and backward as one kernel each
    (yond_film_silu_f32 / yond_film_silu_bwd_f32) instead of autograd's multiply, add, SiLU and their five backward passes.
    z [N][H][W][C]; tk, tb [N][C]."""

    @staticmethod
    def forward(ctx, z, tk, tb, plan):
        z, tk, tb = z.contiguous(), tk.contiguous(), tb.contiguous()
        N, H, W, C = z.shape
        out = torch.empty_like(z)
        L.check(plan.lib.yond_film_silu_f32(L.ptr(z), L.ptr(tk), L.ptr(tb), L.ptr(out), N, H * W, C, L.stream()), "yond_film_silu_f32")
        ctx.save_for_backward(z, tk, tb)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, dout):
        z, tk, tb = ctx.saved_tensors
        dout = dout.contiguous()
        N, H, W, C = z.shape
        dz, dtk, dtb = torch.empty_like(z), torch.empty_like(tk), torch.empty_like(tb)
        L.check(ctx.plan.lib.yond_film_silu_bwd_f32(L.ptr(z), L.ptr(tk), L.ptr(tb), L.ptr(dout), L.ptr(dz), L.ptr(dtk), L.ptr(dtb), N, H * W, C,
                                                    L.stream()), "yond_film_silu_bwd_f32")
        return dz, dtk, dtb, None


class _FilmMLP(torch.autograd.Function):
    """The sigma-conditioning of a guided block (archs/modules.py:170-178: gamma = Conv2d(1, C, 1) -> SiLU -> Conv2d(C, C, 1),
    beta = SiLU -> Conv2d(C, C, 1) on t [B][1][1][1]) as ONE kernel forward and two backward (yond_film_mlp_fwd_f32 / _bwd_f32)
    instead of autograd's ~24 broadcast / reduce launches over [B][C][C] temporaries.  Returns (tk, tb) [B][cp], zero beyond C."""

    @staticmethod
    def forward(ctx, t, w1, b1, w2, b2, w3, b3, plan, cp):
        B, C = t.shape[0], w1.shape[0]
        t = t.contiguous()
        ws = [p if p.is_contiguous() else p.contiguous() for p in (w1, b1, w2, b2, w3, b3)]
        tk = torch.empty((B, cp), dtype=torch.float32, device=t.device)
        tb = torch.empty((B, cp), dtype=torch.float32, device=t.device)
        L.check(plan.lib.yond_film_mlp_fwd_f32(L.ptr(t), *(L.ptr(p) for p in ws), B, C, cp, L.ptr(tk), L.ptr(tb), L.stream()),
                "yond_film_mlp_fwd_f32")
        ctx.save_for_backward(t, ws[0], ws[1], ws[2], ws[4], tk)
        ctx.plan, ctx.shapes = plan, (w1.shape, w2.shape, w3.shape)
        return tk, tb

    @staticmethod
    def backward(ctx, dtk, dtb):
        t, w1, b1, w2, w3, tk = ctx.saved_tensors
        B, C = t.shape[0], w1.shape[0]
        cp = tk.shape[1]
        dev = t.device
        dtk = torch.zeros_like(tk) if dtk is None else dtk.contiguous()
        dtb = torch.zeros_like(tk) if dtb is None else dtb.contiguous()
        scratch = torch.empty(2 * B * C, dtype=torch.float32, device=dev)
        dw1, db1, db2, db3 = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4))
        dW2, dW3 = (torch.empty((C, C), dtype=torch.float32, device=dev) for _ in range(2))
        L.check(ctx.plan.lib.yond_film_mlp_bwd_f32(L.ptr(t), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(w3), L.ptr(tk), L.ptr(dtk), L.ptr(dtb),
                                                   B, C, cp, L.ptr(scratch), L.ptr(dw1), L.ptr(db1), L.ptr(dW2), L.ptr(db2), L.ptr(dW3), L.ptr(db3),
                                                   L.stream()), "yond_film_mlp_bwd_f32")
        s1, s2, s3 = ctx.shapes
        return None, dw1.view(s1), db1, dW2.view(s2), db2, dW3.view(s3), db3, None, None


class _FilmMLPAll(torch.autograd.Function):
    """_FilmMLP for ALL guided blocks of the net in one call: the MLPs depend on sigma and their own weights only, so every block's
    forward runs before the first convolution (2 launches) and every block's backward after the last (5 launches) -- per block they
    were 2 + 5 launches of ~12 us of latency each, 63 per step.  apply(t, plan, cps, *params) with six parameters per block in _FilmMLP's
    order; returns (tk_0, tb_0, tk_1, tb_1, ...), views of one zero-initialised buffer."""

    @staticmethod
    def forward(ctx, t, plan, cps, *params):
        nb = len(cps)
        B = t.shape[0]
        t = t.contiguous()
        params = [q if q.is_contiguous() else q.contiguous() for q in params]
        buf = torch.zeros(2 * B * sum(cps), dtype=torch.float32, device=t.device)
        outs, off = [], 0
        descs = (L.FilmMlpDesc * nb)()
        for i, cp in enumerate(cps):
            w1, b1, w2, b2, w3, b3 = params[6 * i:6 * i + 6]
            tk = buf[off:off + B * cp].view(B, cp)
            tb = buf[off + B * cp:off + 2 * B * cp].view(B, cp)
            off += 2 * B * cp
            d = descs[i]
            d.t, d.w1, d.b1, d.W2, d.b2, d.W3, d.b3 = (q.data_ptr() for q in (t, w1, b1, w2, b2, w3, b3))
            d.tk, d.tb = tk.data_ptr(), tb.data_ptr()
            d.B, d.C, d.ld = B, w1.shape[0], cp
            outs += [tk, tb]
        L.check(plan.lib.yond_film_mlp_fwd_multi_f32(C.cast(descs, C.c_void_p), nb, L.stream()), "yond_film_mlp_fwd_multi_f32")
        ctx.save_for_backward(t, *params, *outs[0::2])       # (outputs go through save_for_backward: a plain attribute would keep the
        ctx.plan, ctx.cps = plan, tuple(cps)                 #  step's autograd graph alive into the next step -- and into a graph capture)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        nb = len(ctx.cps)
        t, *rest = ctx.saved_tensors
        params, tks = rest[:6 * nb], rest[6 * nb:]
        B, dev = t.shape[0], t.device
        Cs = [params[6 * i].shape[0] for i in range(nb)]
        need = sum(2 * B * c + 4 * c + 2 * c * c for c in Cs)
        buf = torch.empty(need, dtype=torch.float32, device=dev)
        descs = (L.FilmMlpDesc * nb)()
        keep, out, off = [], [], 0

        def take(n, shape):
            nonlocal off
            v = buf[off:off + n].view(shape)
            off += n
            return v
        for i, (cp, c) in enumerate(zip(ctx.cps, Cs)):
            w1, b1, w2, b2, w3, b3 = params[6 * i:6 * i + 6]
            tk = tks[i]
            dtk, dtb = grads[2 * i], grads[2 * i + 1]
            dtk = torch.zeros_like(tk) if dtk is None else dtk.contiguous()
            dtb = torch.zeros_like(tk) if dtb is None else dtb.contiguous()
            keep += [dtk, dtb]
            scratch = take(2 * B * c, (2 * B * c,))
            dw1, db1, db2, db3 = take(c, w1.shape), take(c, (c,)), take(c, (c,)), take(c, (c,))
            dW2, dW3 = take(c * c, w2.shape), take(c * c, w3.shape)
            d = descs[i]
            d.t, d.w1, d.b1, d.W2, d.W3 = (q.data_ptr() for q in (t, w1, b1, w2, w3))
            d.tk, d.dtk, d.dtb, d.scratch = tk.data_ptr(), dtk.data_ptr(), dtb.data_ptr(), scratch.data_ptr()
            d.dw1, d.db1, d.dW2, d.db2, d.dW3, d.db3 = (q.data_ptr() for q in (dw1, db1, dW2, db2, dW3, db3))
            d.B, d.C, d.ld = B, c, cp
            out += [dw1, db1, dW2, db2, dW3, db3]
        L.check(ctx.plan.lib.yond_film_mlp_bwd_multi_f32(C.cast(descs, C.c_void_p), nb, L.stream()), "yond_film_mlp_bwd_multi_f32")
        return (None, None, None) + tuple(out)


class _SiluRes(torch.autograd.Function):
    """A guided block's input: returns (SiLU(x), x) -- conv1's operand and the tensor the block adds to its output.  Backward joins
    the two gradients in one pass, dx = dres + dz SiLU'(x) (yond_silu_bwd_add_f32), where autograd ran silu_backward and then an
    accumulation kernel."""

    @staticmethod
    def forward(ctx, x, plan):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.plan = plan
        if x.numel() % 4:
            return F.silu(x), x.view_as(x)
        z = torch.empty_like(x)
        L.check(plan.lib.yond_silu_f32(L.ptr(x), L.ptr(z), x.numel(), L.stream()), "yond_silu_f32")
        return z, x.view_as(x)

    @staticmethod
    def backward(ctx, dz, dres):
        x, = ctx.saved_tensors
        if dz is None or dres is None:
            g = dres if dz is None else dz * (torch.sigmoid(x) * (1 + x * (1 - torch.sigmoid(x))))
            return g, None
        dz, dres = dz.contiguous(), dres.contiguous()
        dx = torch.empty_like(x)
        L.check(ctx.plan.lib.yond_silu_bwd_add_f32(L.ptr(x), L.ptr(dz), L.ptr(dres), L.ptr(dx), x.numel(), L.stream()), "yond_silu_bwd_add_f32")
        return dx, None


def ctx_splits(w, x0, x1):
    """Real channel counts of the two sources of a two-source 1x1 layer: GuidedResUnet's decoder concatenates `up` (c) and the
    skip tensor (c) into 2c channels (archs/Unet.py:447-461)."""
    c = w.shape[1] // 2
    return c, w.shape[1] - c


class _ConvT2x2(torch.autograd.Function):
    """nn.ConvTranspose2d(cin, cout, 2, stride=2): x [N][H][W][cin_p] -> [N][2H][2W][cout_p]."""

    @staticmethod
    def forward(ctx, x, w, b, plan):
        N, H, W, _ = x.shape
        x = x.contiguous()
        if _use_gemm(plan, x.shape[-1]):                     # a GEMM [P][cin] x [cin][4 cout] with a pixel-shuffle store
            cin, cout = w.shape[0], w.shape[1]
            cop = _rup(cout)
            y = torch.empty((N, 2 * H, 2 * W, cop), dtype=torch.float32, device=x.device)
            _gemm_split(plan, [(x, w.data_ptr(), 4 * cout, 0, x.shape[-1], x.shape[-1], _BIG, cin)], N * H * W, 4 * cop, cout, 4, 1, cop, b, y, cop,
                        1, H, W)
        else:
            y = _conv_fwd(plan, w, b, 1, 1, [w.shape[0]], [x], N, H, W, shuffle=True)
        ctx.save_for_backward(x, w)
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        plan = ctx.plan
        dy = dy.contiguous()
        N, H, W, cin_p = x.shape
        cin, cout = w.shape[0], w.shape[1]
        cop = dy.shape[-1]
        with _lane(plan):
            dw = _wgrad(plan, x, dy, 1, 2, 4)[:, :cout, :cin].permute(2, 1, 0).reshape(cin, cout, 2, 2)     # [tap][co][ci] -> [ci][co][dy][dx]
            db = _colsum(plan, dy)[:cout]
        # adjoint: dx[y][x][ci] = sum_{dy,dx,co} g[2y+dy][2x+dx][co] W[ci][co][dy][dx] = a 1x1 GEMM on the pixel-unshuffled gradient
        gu = dy.reshape(N, H, 2, W, 2, cop).permute(0, 1, 3, 2, 4, 5).reshape(N, H, W, 4 * cop).contiguous()

        def xf(t):
            m = torch.zeros((cin, 4, cop), dtype=torch.float32, device=t.device)
            m[:, :, :cout] = t.permute(0, 2, 3, 1).reshape(cin, 4, cout)
            return m.reshape(cin, 4 * cop, 1, 1)
        if _use_gemm(plan, 4 * cop):                         # dx[p][ci] = sum_{j, co} gu[p][j cop + co] w[ci][co][j]
            dx = torch.empty((N, H, W, cin_p), dtype=torch.float32, device=dy.device)
            _gemm_split(plan, [(gu, w.data_ptr(), 4, 1, 4 * cop, 4 * cop, cop, cout)], N * H * W, cin_p, cin, 4 * cout, 0, _BIG, None, dx, cin_p)
            _join(plan)
            return dx, dw, db, None
        guf = _flat32(gu)                                    # (a 1x1 GEMM: narrow images as 32-pixel rows)
        dx = _conv_fwd(plan, w, None, 1, 1, [4 * cop], [guf], guf.shape[0], guf.shape[1], guf.shape[2], role='dgrad', xf=xf)
        dx = dx if dx.shape[-1] == cin_p else _pad_c(dx[..., :cin], cin_p)
        _join(plan)
        return dx.view(N, H, W, dx.shape[-1]), dw, db, None


class TrainStep:
    """One optimisation step of a yond_public_amd.archs.GuidedResUnet or UNetSeeInDark (parameter names / shapes of the reference)."""

    def __init__(self, module, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, charbonnier=False, ddp=None, conv='split', loss_scale=None, graph=True):
        """conv: 'split' -- forward and data-gradient 3x3 convolutions as fp32-accurate split-operand products on the fp16 matrix
        cores (the inference path's kernels; 22-bit operands, fp32 accumulation); 'fp32' -- every convolution on the fp32-input
        MFMA kernels (exact fp32 products).
        loss_scale: the split-operand kernels stage every operand as two fp16 halves, which is fp32-accurate only while the
        operand is in fp16's NORMAL range (|a| >= 6.1e-5); the back-propagated values of a mean loss over n = 4.2 M elements
        (the reference's batch: dpred = +-1/n = 2.4e-7) are far below it.  So dpred is multiplied by a power of two S before
        backward() and the flat gradient by 1/S behind the (data-parallel) reduction, in front of Adam -- both exact in float32; None: S = the power of
        two that puts S/n into [1/8, 1/4) (2^ceil(log2 n) / 8); 1: no scaling.  Under data parallelism the buckets are all-reduced at scale S and
        unscaled afterwards, so S must be the same on every rank (equal local batch sizes; the overflow history is shared because the retry
        decision is collective): checked collectively the first time a batch shape is seen.  A gradient or activation that leaves fp16's range (|a| > 65504)
        is reported by the kernels' status word, read once per step: the step is redone at S/256, twice at most.  charbonnier: Unet_Loss(charbonnier=True) (losses/base_loss.py:82-85).  ddp: None -- average the gradients over the
        ranks whenever a process group exists (the reference wraps the net in DDP whenever it sees more than one GPU,
        trainer_AWGN.py:59-61); False -- never.
        graph: replay the step (gather, forward, loss, backward, unscale, Adam: ~500 launches) as ONE hipGraph from the third step of
        a given batch shape on -- single process only (a gradient all-reduce inside a captured region is not attempted).  The
        inputs are copied into the graph's fixed buffers; the returned gradients and `last_pred` are then the graph's own buffers,
        overwritten by the next step."""
        from . import distributed as D
        self.graph = bool(graph)
        self._graphs, self._seen = {}, {}
        self.m = module
        self.dev = next(module.parameters()).device
        self.plan = _plan(self.dev)
        if conv not in ('split', 'fp32'):
            raise ValueError(f"conv must be 'split' or 'fp32', got {conv!r}")
        self.plan.train_conv = conv
        self.loss_scale = loss_scale
        self.lr, self.betas, self.eps = lr, betas, eps
        self.charbonnier = bool(charbonnier)
        self.params = dict(module.named_parameters())
        # the parameter arena: [0] a constant zero, then every parameter back to back; the Parameters become views of it (Adam
        # is ONE launch over the arena, and every convolution's packed weights are a gather of it: _Packing)
        n = sum(p.numel() for p in self.params.values())
        self.arena = torch.zeros(n + 1, dtype=torch.float32, device=self.dev)
        off = 1
        self.slots = {}
        for k, p in self.params.items():
            if p.dtype != torch.float32 or p.device != self.arena.device:
                raise L.YondHipError(f"TrainStep: parameter {k} must be a float32 tensor on {self.dev}")
            self.arena[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.arena[off:off + p.numel()].view(p.shape)
            self.slots[k] = (off, p.numel())
            off += p.numel()
        self.plan.arena = self.arena
        self.adam_m, self.adam_v = torch.zeros_like(self.arena), torch.zeros_like(self.arena)
        self.state = {k: (self.adam_m[o:o + c].view(self.params[k].shape), self.adam_v[o:o + c].view(self.params[k].shape))
                      for k, (o, c) in self.slots.items()}
        self.t = 0
        self.reducer = None
        if ddp is not False and D._active():
            D.broadcast_params(self.params.values(), src=0)         # DDP's constructor: every rank starts from rank 0's weights
            self.reducer = D.GradReducer(self.params.values())

    # -- forward on padded NHWC tensors ------------------------------------------------------------------------------
    def _block(self, pre, x, xs, t, cp):
        P = self.params
        if xs is not None:                                   # decoder: short_cut = 1x1 over cat(up, skip)
            x = _Conv1x1.apply(x, xs, P[pre + '.short_cut.0.weight'], P[pre + '.short_cut.0.bias'], self.plan)
        c = P[pre + '.conv1.weight'].shape[0]
        # gamma / beta: 1x1 convolutions on a (B, 1, 1, 1) tensor = three tiny linear layers (archs/modules.py:170-178), one kernel
        if isinstance(t, dict):                              # every block's MLPs at once (forward: _FilmMLPAll)
            tk, tb = t[pre]
        else:
            tk, tb = _FilmMLP.apply(t, P[pre + '.gamma.0.weight'], P[pre + '.gamma.0.bias'], P[pre + '.gamma.2.weight'],
                                    P[pre + '.gamma.2.bias'], P[pre + '.beta.1.weight'], P[pre + '.beta.1.bias'], self.plan, cp)
        z, xr = _SiluRes.apply(x, self.plan)                 # SiLU(x) for conv1, x itself for the residual (gradients joined in one pass)
        z = _Conv3x3.apply(z, P[pre + '.conv1.weight'], P[pre + '.conv1.bias'], self.plan, 1, True)
        if self.plan.lib.yond_film_silu_supported(cp):
            z = _FilmSilu.apply(z, tk, tb, self.plan)
        else:
            z = F.silu(z * tk[:, None, None, :] + tb[:, None, None, :])
        # z = conv2(z) + x with the residual in the convolution's epilogue (archs/modules.py:194-195)
        return _Conv3x3.apply(z, P[pre + '.conv2.weight'], P[pre + '.conv2.bias'], self.plan, 1, True, xr)

    def _block_snr(self, pre, x, xs, t, cp):
        """SNR_Block (archs/modules.py:198-233): z = conv2(SiLU(conv1(SiLU(x)) * a1)) * a2 + x with two multiplicative gates
        a = sfm(t) = W2 . SiLU(W1 t + b1) + b2 (1x1 convolutions on a (B, 1, 1, 1) tensor: tiny linear layers, autograd's own).  The 3x3
        convolutions and the SiLU-with-scale kernel are the guided block's; the second gate and the residual are elementwise glue."""
        P = self.params
        if xs is not None:
            x = _Conv1x1.apply(x, xs, P[pre + '.short_cut.0.weight'], P[pre + '.short_cut.0.bias'], self.plan)
        c = P[pre + '.conv1.weight'].shape[0]

        def gate(name):
            w1, b1 = P[f'{pre}.{name}.0.weight'].reshape(c), P[f'{pre}.{name}.0.bias']
            w2, b2 = P[f'{pre}.{name}.2.weight'].reshape(c, c), P[f'{pre}.{name}.2.bias']
            a = F.linear(F.silu(t[:, None] * w1[None, :] + b1[None, :]), w2, b2)          # [B][c]
            return _pad_c(a, cp) if cp != c else a
        a1, a2 = gate('sfm1'), gate('sfm2')
        z, xr = _SiluRes.apply(x, self.plan)
        z = _Conv3x3.apply(z, P[pre + '.conv1.weight'], P[pre + '.conv1.bias'], self.plan, 1, True)
        if self.plan.lib.yond_film_silu_supported(cp):
            z = _FilmSilu.apply(z, a1, torch.zeros_like(a1), self.plan)
        else:
            z = F.silu(z * a1[:, None, None, :])
        z = _Conv3x3.apply(z, P[pre + '.conv2.weight'], P[pre + '.conv2.bias'], self.plan, 1, True)
        return z * a2[:, None, None, :] + xr

    def _forward_unet(self, x_nchw):
        """UNetSeeInDark (archs/Unet.py:55-104): conv -> LeakyReLU(0.2) pairs, 2x2 max pooling, ConvTranspose2d + cat + conv; the
        pooling and the activations are autograd's elementwise glue, every convolution runs on the HIP kernels."""
        m, P = self.m, self.params
        x = x_nchw.permute(0, 2, 3, 1).contiguous()
        B = x.shape[0]
        ub = None
        if m.norm:
            ub = x.reshape(B, -1).max(1).values.detach()
            x = x / ub[:, None, None, None]
        act = lambda z: F.leaky_relu(z, 0.2)
        pool = lambda z: F.max_pool2d(z.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).contiguous()
        cur = act(_Conv3x3.apply(_pad_c(x, 32), P['conv1_1.weight'], P['conv1_1.bias'], self.plan, 1, False))
        cur = act(_Conv3x3.apply(cur, P['conv1_2.weight'], P['conv1_2.bias'], self.plan, 1, True))
        skips = {1: cur}
        for i in range(2, 6):
            cur = pool(cur)
            cur = act(_Conv3x3.apply(cur, P[f'conv{i}_1.weight'], P[f'conv{i}_1.bias'], self.plan, 1, True))
            cur = act(_Conv3x3.apply(cur, P[f'conv{i}_2.weight'], P[f'conv{i}_2.bias'], self.plan, 1, True))
            if i < 5:
                skips[i] = cur
        for i in range(6, 10):
            up = _ConvT2x2.apply(cur, P[f'upv{i}.weight'], P[f'upv{i}.bias'], self.plan)
            c0 = P[f'upv{i}.weight'].shape[1]
            cur = act(_Conv3x3Cat.apply(up, skips[10 - i], P[f'conv{i}_1.weight'], P[f'conv{i}_1.bias'], self.plan, c0))
            cur = act(_Conv3x3.apply(cur, P[f'conv{i}_2.weight'], P[f'conv{i}_2.bias'], self.plan, 1, True))
        out = _Conv1x1.apply(cur, None, P['conv10_1.weight'], P['conv10_1.bias'], self.plan)[..., :4]
        if m.res:
            out = out + x[..., :4]
        if m.norm:
            out = out * ub[:, None, None, None]
        return out.permute(0, 3, 1, 2)

    def forward(self, x_nchw, sigma=None):
        m, P = self.m, self.params
        if type(m).__name__ == 'UNetSeeInDark':
            return self._forward_unet(x_nchw)
        snr = type(m).__name__ == 'SNRnet'                   # the same skeleton with SNR_Blocks (archs/Unet.py:288-378)
        if type(m).__name__ != 'GuidedResUnet' and not snr:
            raise NotImplementedError(f"TrainStep: {type(m).__name__} (GuidedResUnet, SNRnet and UNetSeeInDark are built)")
        block = self._block_snr if snr else self._block
        x = x_nchw.permute(0, 2, 3, 1).contiguous()
        B = x.shape[0]
        t = sigma.reshape(B).to(torch.float32)
        ub = None
        if m.norm:                                           # data_normalize: the per-image maximum is a constant (modules.py:16-20)
            ub = x.reshape(B, -1).max(1).values.detach()
            x = x / ub[:, None, None, None]
            t = t / ub
        nf = P['conv_in.weight'].shape[0]
        if FILM_ALL and self.reducer is None and not snr:    # the sigma-MLPs of all nine blocks up front (they depend on t only); not under DDP:
                                                             # their gradients would all land last and hold back every bucket's all-reduce
            pres = [f'conv{i}' for i in range(1, 10)]
            cps = [_rup(nf * 2 ** (min(i, 10 - i) - 1)) for i in range(1, 10)]
            flat = [P[pre + k] for pre in pres for k in ('.gamma.0.weight', '.gamma.0.bias', '.gamma.2.weight', '.gamma.2.bias',
                                                         '.beta.1.weight', '.beta.1.bias')]
            outs = _FilmMLPAll.apply(t, self.plan, cps, *flat)
            t = {pre: (outs[2 * i], outs[2 * i + 1]) for i, pre in enumerate(pres)}
        a = _Conv3x3.apply(_pad_c(x, 32), P['conv_in.weight'], P['conv_in.bias'], self.plan, 1, False)
        cur = F.leaky_relu(a, 0.01)
        skips = {}
        for i in range(1, 5):
            cp = _rup(nf * 2 ** (i - 1))
            cur = block(f'conv{i}', cur, None, t, cp)
            skips[i] = cur
            cur = _Conv3x3.apply(cur, P[f'pool{i}.conv.weight'], P[f'pool{i}.conv.bias'], self.plan, 2, True)
        cur = block('conv5', cur, None, t, _rup(nf * 16))
        for i in range(6, 10):
            up = _ConvT2x2.apply(cur, P[f'upv{i}.weight'], P[f'upv{i}.bias'], self.plan)
            cur = block(f'conv{i}', up, skips[10 - i], t, _rup(nf * 2 ** (9 - i)))
        out = _Conv1x1.apply(cur, None, P['conv10.weight'], P['conv10.bias'], self.plan)[..., :4]
        if m.res:
            out = out + x[..., :4]
        if m.norm:
            out = out * ub[:, None, None, None]
        return out.permute(0, 3, 1, 2)

    def _gather_weights(self):
        """ONE gather for the maps of every convolution met so far (forward and data-gradient roles, biases): ~100 small launches
        per step otherwise.  Layers first met during this step gather on their own and join the batch at the next step."""
        plan = self.plan
        packs = list(plan.wcache.values())
        if not packs:
            plan.wbatch = None
            return
        if getattr(self, '_wb_count', -1) != len(packs):
            slots, maps, off = {}, [], 0
            for pk in packs:
                for m in pk.maps():
                    slots[id(m)] = (off, m.numel())
                    maps.append(m)
                    off += m.numel()
            plan.gen += 1
            self._wb_map, self._wb_slots, self._wb_count = torch.cat(maps), slots, len(packs)
            self._wb_buf = torch.empty(off, dtype=torch.float32, device=self.dev)
            # the split-operand layers' packing as ONE launch: descriptor table + one destination buffer whose slices the layers keep
            desc, off_g, off_d, self._pk_ids = [], 0, 0, set()
            for pk in packs:
                if pk.split_tn:
                    so, n = slots[id(pk.omap)]
                    taps = pk.pc.ksize * pk.pc.ksize
                    desc.append([so, pk.pc.gemm_n, pk.pc.cinp, taps, pk.split_tn, off_d, off_g])
                    off_g += n // 4
                    off_d += n
                    self._pk_ids.add(id(pk))
            self._pk_groups = off_g
            if desc:
                self._pk_desc = torch.tensor(desc, dtype=torch.int64, device=self.dev)
                self._pk_buf = torch.empty(off_d, dtype=torch.float32, device=self.dev)
                for pk, dsc in zip([q for q in packs if q.split_tn], desc):
                    pk.pc._packed[('split', 2)] = (pk.split_tn, self._pk_buf[dsc[5]:dsc[5] + dsc[1] * dsc[2] * dsc[3]])
        torch.index_select(self.arena, 0, self._wb_map, out=self._wb_buf)
        if self._pk_groups:
            L.check(plan.lib.yond_pack_conv_split_weights_batch_dev_f32(L.ptr(self._wb_buf), L.ptr(self._pk_desc), self._pk_desc.shape[0],
                                                                        L.ptr(self._pk_buf), self._pk_groups, L.ptr(plan.status[1:2]), L.stream()),
                    "yond_pack_conv_split_weights_batch_dev_f32")
        plan.wbatch = (self._wb_buf, self._wb_slots, self._pk_ids)

    def _gemm_fallback(self):
        """Move the plan's 1x1 / transposed layers from the split-operand GEMM back to the fp32-input MFMA path (which has no operand
        range limit), once: True if that changed anything.  Called when a range trip survives every loss scale -- then it is a forward
        activation, and the GEMM's x operand is the one such tensor no 3x3 kernel had looked at before."""
        plan = self.plan
        if not (GEMM_SPLIT and not getattr(plan, 'gemm_off', False) and getattr(plan, 'train_conv', 'split') == 'split'):
            return False
        import warnings
        warnings.warn("TrainStep: a range trip at every loss scale: the split-operand GEMMs are switched off for this plan (fp32-input MFMA instead)")
        plan.gemm_off = True
        self._graphs.clear()
        self._seen.clear()
        plan.status.zero_()
        return True

    def _weight_trip(self):
        """Status word 1: a weight beyond what a split-operand kernel can stage.  The 3x3 kernels and the weight packers take |w| <= 65504;
        the GEMM (1x1 / transposed layers, csrc/gemm_split.hip) stages 2^11 w and therefore needs |w| < 32 -- a limit the reference does
        not have.  The first trip moves those layers back to the fp32-input MFMA path for this plan (and drops the captured graphs, which
        hold GEMM launches); a trip without GEMMs in the step is a weight no fp16 part can hold."""
        plan = self.plan
        if GEMM_SPLIT and not getattr(plan, 'gemm_off', False) and getattr(plan, 'train_conv', 'split') == 'split':
            plan.gemm_off = True
            self._graphs.clear()
            self._seen.clear()
            plan.status.zero_()
            return
        raise L.YondHipError("TrainStep: a weight is NaN or beyond the split-operand kernels' range (|w| > 65504; |w| >= 32 in a 1x1 / "
                             "transposed layer on the split GEMM): TrainStep(conv='fp32') keeps every convolution on the fp32-input MFMA")

    # -- loss, backward, Adam ----------------------------------------------------------------------------------------
    def _scale_for(self, n):
        """The step's loss scale: the fixed one, or the automatic one -- S / n in [1/8, 1/4), lowered (and kept lower) whenever
        a step overflowed, raised again by a factor of two after 500 clean steps (the usual dynamic loss scaling)."""
        if self.loss_scale is not None:
            return float(self.loss_scale)
        if self.plan.train_conv != 'split':
            return 1.0
        top = 2.0 ** (math.ceil(math.log2(max(n, 1))) - 3)
        cur = getattr(self, '_auto_scale', None)
        if cur is None or cur > top:
            cur = top
        elif cur < top and getattr(self, '_clean_steps', 0) >= 500:
            cur, self._clean_steps = cur * 2.0, 0
        self._auto_scale = cur
        return cur

    def _fwd_bwd(self, imgs_lr, imgs_hr, sigma, S):
        """Forward, loss, backward with dpred scaled by S.  Returns (pred, loss sum [1] float64 on the device)."""
        lib = self.plan.lib
        for p in self.params.values():
            p.grad = None
        if self.reducer is not None:
            self.reducer.begin()
        self.plan.status.zero_()
        self._gather_weights()
        pred = self.forward(imgs_lr, sigma).contiguous()
        tgt = imgs_hr.contiguous()
        loss_sum = torch.zeros(1, dtype=torch.float64, device=self.dev)
        dpred = torch.empty_like(pred)
        if self.charbonnier:
            L.check(lib.yond_charbonnier_loss_f32(L.ptr(pred), L.ptr(tgt), pred.numel(), 1e-6, L.ptr(loss_sum), L.ptr(dpred), L.stream()),
                    "yond_charbonnier_loss_f32")
        else:
            L.check(lib.yond_l1_loss_f32(L.ptr(pred), L.ptr(tgt), pred.numel(), L.ptr(loss_sum), L.ptr(dpred), L.stream()), "yond_l1_loss_f32")
        if S != 1.0:
            dpred.mul_(S)                                    # (a power of two: exact)
        pred.backward(dpred)                                 # (the reducer's hooks launch a bucket's all-reduce as its last gradient lands)
        return pred, loss_sum

    # -- the step as one hipGraph ------------------------------------------------------------------------------------
    def _hyp(self, t):
        """The two scalars of Adam step t, in float64 as yond_adam_step_f32 computes them (csrc/train.hip)."""
        return [self.lr / (1.0 - self.betas[0] ** t), 1.0 / math.sqrt(1.0 - self.betas[1] ** t)]

    def _flat_grad(self, S):
        zero = self.arena.new_zeros(1)
        flat = torch.cat([zero] + [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params.values()])
        if S != 1.0:
            flat.mul_(1.0 / S)                               # exact
        return flat

    def _capture(self, imgs_lr, imgs_hr, sigma, S):
        """Capture gather + forward + loss + backward + unscale + Adam at loss scale S for this batch shape.  Adam reads its step
        scalars from g.hyp and does nothing when the step's status words report a range trip or a non-finite loss (word 2), so a
        replay never writes a bad update: the host then redoes the step at a lower scale, as the eager path does."""
        g = types.SimpleNamespace(S=S)
        g.lr_in, g.hr_in = imgs_lr.detach().clone(), imgs_hr.detach().clone()
        g.sigma = None if sigma is None else sigma.detach().clone()
        g.hyp = torch.zeros(2, dtype=torch.float32, device=self.dev)
        lib = self.plan.lib
        g.graph = torch.cuda.CUDAGraph()
        self._gather_weights()                               # (tables and buffers for every layer met so far are built HERE, not while capturing:
        torch.cuda.synchronize(self.dev)                     #  a host-to-device copy inside a capture is not allowed)
        gen0 = self.plan.gen
        with torch.cuda.graph(g.graph):
            g.pred, g.loss_sum = self._fwd_bwd(g.lr_in, g.hr_in, g.sigma, S)
            self.plan.status[2:3].copy_(torch.logical_not(torch.isfinite(g.loss_sum)).to(torch.int32))
            g.flat = self._flat_grad(S)
            L.check(lib.yond_adam_step_dev_f32(L.ptr(self.arena), L.ptr(g.flat), L.ptr(self.adam_m), L.ptr(self.adam_v), self.arena.numel(),
                                               self.betas[0], self.betas[1], self.eps, L.ptr(g.hyp), L.ptr(self.plan.status), L.stream()),
                    "yond_adam_step_dev_f32")
        g.gen = self.plan.gen if self.plan.gen == gen0 else -1     # (a buffer replaced DURING the capture: use this graph once at most)
        return g

    def _graph_step(self, key, imgs_lr, imgs_hr, sigma, S):
        """One replay.  Returns (loss, grads) or None when the step tripped the range guard (no update was applied)."""
        g = self._graphs.get(key)
        if g is not None and g.gen != self.plan.gen:         # a workspace / packed-weight buffer its launches point at has been replaced since
            del self._graphs[key]                            # (another batch shape met new layers): capture again
            g = None
        if g is None:
            while len(self._graphs) >= 4:                    # (every graph keeps its own activations: a handful of batch shapes at most)
                del self._graphs[next(iter(self._graphs))]
            g = self._graphs[key] = self._capture(imgs_lr, imgs_hr, sigma, S)
        g.lr_in.copy_(imgs_lr)
        g.hr_in.copy_(imgs_hr)
        if sigma is not None:
            g.sigma.copy_(sigma)
        g.hyp.copy_(torch.tensor(self._hyp(self.t + 1), dtype=torch.float64).to(torch.float32), non_blocking=False)
        g.graph.replay()
        st = self.plan.status.cpu()                          # (the step's synchronisation point)
        if int(st[1]) & 1:
            self._weight_trip()                              # (raises unless the GEMM's narrower limit can be lifted)
            return None                                      # the eager path redoes the step without the split-operand GEMMs
        if int(st[0]) & 1:
            return None
        loss = float(g.loss_sum.item()) / g.pred.numel()
        if not math.isfinite(loss):
            raise L.YondHipError(f"TrainStep: the loss is {loss}: no update applied")
        self.t += 1
        self.last_scale = S
        self._clean_steps = getattr(self, '_clean_steps', 0) + 1
        self.last_pred = g.pred.detach()
        self.plan.wbatch = None
        self.m._plan = None
        return loss, {k: g.flat[o:o + c].view(self.params[k].shape) for k, (o, c) in self.slots.items()}

    # -- loss, backward, Adam ----------------------------------------------------------------------------------------
    def step(self, imgs_lr, imgs_hr, sigma=None):
        """trainer_AWGN.py:101-117 for one batch (`pred = net(imgs_lr, sigma)` for a guided net, `net(imgs_lr)` otherwise).
        Returns (loss, {name: gradient})."""
        lib = self.plan.lib
        n_out = imgs_hr.numel()
        S = self._scale_for(n_out)
        if self.graph and self.reducer is None and self.dev.type == 'cuda':
            key = (tuple(imgs_lr.shape), tuple(imgs_hr.shape), None if sigma is None else tuple(sigma.shape), S)
            self._seen[key] = self._seen.get(key, 0) + 1
            if self._seen[key] > 2:                          # (two eager steps first: every lazily built buffer exists by then)
                out = self._graph_step(key, imgs_lr, imgs_hr, sigma, S)
                if out is not None:
                    return out                               # (else: a range trip, nothing updated -- the eager path lowers the scale)
        if self.reducer is not None:
            # the ranks all-reduce S-scaled gradients: S must agree (see __init__).  Checked EVERY step, by every rank alike: the decision to
            # check is itself collective (a check keyed on this rank's own batch shapes would be issued by one rank and skipped by another --
            # exactly when their shapes differ, the case it exists for -- and mismatch the gradient all-reduces that follow)
            import torch.distributed as dist
            t = torch.tensor([S, -S], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if float(t[0]) != S or float(t[1]) != -S:
                raise L.YondHipError(f"TrainStep: the ranks' loss scales differ (this rank {S:g}, maximum {float(t[0]):g}, minimum {-float(t[1]):g}): "
                                     "give every rank the same local batch size or a fixed loss_scale")
        S0, auto0, clean0 = S, getattr(self, '_auto_scale', None), getattr(self, '_clean_steps', 0)    # the scale this step started with
        attempt = 0
        while True:
            pred, loss_sum = self._fwd_bwd(imgs_lr, imgs_hr, sigma, S)
            if self.reducer is not None:
                # every rank must take the same branch below (a retry re-issues the gradient all-reduces): the ranks' words OR-ed
                import torch.distributed as dist
                dist.all_reduce(self.plan.status, op=dist.ReduceOp.MAX)
            st = self.plan.status.cpu()                      # ONE read per step (it also is the step's synchronisation point)
            if int(st[1]) & 1:
                self._weight_trip()                          # (raises the second time: at most one such redo per plan)
                if self.reducer is not None:
                    self.reducer.finish()
                continue                                     # redone without the split-operand GEMMs; nothing has been applied
            if not (int(st[0]) & 1):
                break
            if attempt == 2 or S == 1.0 and self.loss_scale is not None:
                if self._gemm_fallback():
                    # an x operand of a split GEMM (a 1x1 / transposed layer's input, e.g. the ConvTranspose output that only the 1x1
                    # shortcut reads) is what cannot be lowered by the loss scale: those layers go back to the fp32-input MFMA
                    if self.reducer is not None:
                        self.reducer.finish()
                    # ... and the trip was no gradient overflow: the retries' lowered scale (kept in _auto_scale, regrown x2 per 500 clean steps)
                    # would leave dpred = S / n below fp16's normal range for thousands of steps -- back to the scale the step started with
                    S = S0
                    if self.loss_scale is None:
                        self._auto_scale, self._clean_steps = auto0, clean0
                    attempt = 0
                    continue
                raise L.YondHipError("TrainStep: an activation or gradient is NaN or left fp16's range (|a| > 65504) in the split-operand "
                                     f"kernels (loss scale {S:g}); TrainStep(conv='fp32') has no such limit")
            if self.reducer is not None:
                self.reducer.finish()                        # (drain the all-reduces of the abandoned attempt)
            attempt += 1
            S = max(S / 256.0, 1.0)                          # a gradient overflowed at this scale: redo the step lower
            if self.loss_scale is None:
                self._auto_scale, self._clean_steps = S, 0   # ... and keep the lower scale for the steps that follow
        self.last_scale = S
        self._clean_steps = getattr(self, '_clean_steps', 0) + 1
        self.last_pred = pred.detach()                       # the trainer's running PSNR (trainer_AWGN.py:120-124)
        if self.reducer is not None:
            self.reducer.finish()                            # .grad = the mean over the ranks
        self.t += 1
        flat = self._flat_grad(S)
        grads = {k: flat[o:o + c].view(self.params[k].shape) for k, (o, c) in self.slots.items()}
        if self.params[next(iter(self.params))].data_ptr() != self.arena.data_ptr() + 4:
            raise L.YondHipError("TrainStep: the module's parameters were moved (.to / .float) after the step object was built")
        loss = float(loss_sum.item()) / pred.numel()
        if not math.isfinite(loss):
            raise L.YondHipError(f"TrainStep: the loss is {loss}: no update applied")
        L.check(lib.yond_adam_step_f32(L.ptr(self.arena), L.ptr(flat), L.ptr(self.adam_m), L.ptr(self.adam_v), self.arena.numel(),
                                       self.lr, self.betas[0], self.betas[1], self.eps, self.t, L.stream()), "yond_adam_step_f32")
        self.plan.wbatch = None                              # (the gathered weights are those of before this update)
        self.m._plan = None                                  # the inference plan's packed weights are stale now
        return loss, grads


# -- learning-rate schedule (trainer_base.py:34-46, 138-167) -------------------------------------------------------------
def get_cos_lr(step, period=1000, peak=20, lr=1e-4, ratio=0.4, coldstart=False):
    """WarmUpCosine (trainer_base.py:148-157): linear warm-up over `peak` steps (skipped in the first period of a cold start),
    cosine decay to `ratio`, the whole halved every period."""
    T = step // period
    decay = 2 ** T
    step = step % period
    if step <= peak and (not coldstart or T > 0):
        mul = step / peak
    else:
        mul = (1 - ratio) * (np.cos((step - peak) / (period - peak) * math.pi) * 0.5 + 0.5) + ratio
    return lr * mul / decay


def get_multistep_lr(step, period=1000, lr=1e-4, milestone=(500, 900), gamma=(0.5, 0.1), decay_base=1):
    """trainer_base.py:159-167."""
    decay = decay_base ** (step // period)
    step = step % period
    mul = 1
    for i in range(len(milestone), 0, -1):
        if step > milestone[i - 1]:
            mul = gamma[i - 1]
            break
    return lr * mul / decay


def lr_lambda(hyper):
    """Base_Trainer.get_lr_lambda_func (trainer_base.py:34-46) on the runfile's `hyper` block."""
    num_of_epochs = hyper['stop_epoch'] - hyper['last_epoch']
    step_size = hyper['step_size']
    T = hyper['T'] if 'T' in hyper else 1
    coldstart = True if 'coldstart' not in hyper else hyper['coldstart']
    kind = hyper['lr_scheduler'].lower()
    if 'cos' in kind:
        return lambda x: get_cos_lr(x, period=num_of_epochs // T, lr=hyper['learning_rate'], peak=step_size, coldstart=coldstart)
    if 'multi' in kind:
        return lambda x: get_multistep_lr(x, period=num_of_epochs // T, decay_base=1, milestone=[step_size, step_size * 9 // 5],
                                          gamma=[0.5, 0.1], lr=hyper['learning_rate'])
    raise NotImplementedError(hyper['lr_scheduler'])


class LambdaScheduler:
    """trainer_base.py:138-146: torch's LambdaLR with get_lr = lmbda(last_epoch) -- the lambda returns the learning rate ITSELF,
    not a factor of the optimiser's base rate.  Constructed at last_epoch 0; step() once per epoch (trainer_AWGN.py:153)."""

    def __init__(self, train_step, lmbda):
        self.ts, self.lmbda, self.last_epoch = train_step, lmbda, 0
        self.ts.lr = float(self.lmbda(0))

    def get_last_lr(self):
        return [self.ts.lr]

    def step(self):
        self.last_epoch += 1
        self.ts.lr = float(self.lmbda(self.last_epoch))


class Trainer:
    """The epoch loop of trainer_AWGN.py:78-155 over in-memory batches (data synthesis, checkpoints and plots stay out of scope):
    per epoch every batch is one TrainStep.step, then scheduler.step(); with a process group every rank runs its own batches and
    the gradients are averaged (DDP)."""

    def __init__(self, module, hyper, charbonnier=False, ddp=None):
        self.hyper = dict(hyper)
        self.ts = TrainStep(module, lr=hyper['learning_rate'], charbonnier=charbonnier, ddp=ddp)
        self.scheduler = LambdaScheduler(self.ts, lr_lambda(self.hyper))
        self.history = []                                    # (epoch, learning rate, [losses])

    def train(self, batches, epochs=None):
        """batches: callable(epoch) -> iterable of (imgs_lr, imgs_hr, sigma) for THIS rank."""
        from . import distributed as D
        first = self.hyper['last_epoch'] + 1
        last = self.hyper['stop_epoch'] if epochs is None else first + epochs - 1
        for epoch in range(first, last + 1):
            D.barrier()                                      # trainer_AWGN.py:85
            lr = self.scheduler.get_last_lr()[0]
            losses = [self.ts.step(*b)[0] for b in batches(epoch)]
            self.scheduler.step()
            D.barrier()                                      # :155
            self.history.append((epoch, lr, losses))
        return self.history
