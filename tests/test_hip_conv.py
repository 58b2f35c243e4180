"""GPU: every variant of the convolution kernels behind YondConvDesc -- K2s split-operand products on the fp16 MFMA (the
default path), K2w Winograd / K2 direct on the fp32-input MFMA, the fp16 path, first / last layers, pooling, FiLM --
against torch CPU convolutions (float64 reference of the same op), through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hip_common import report

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def run_conv(w, b, ksize, stride, splits, xs, N, H, W, shuffle=False, **kw):
    from yond_public_amd.engine import _PackedConv, DenoiserPlan
    kw.setdefault('algo', 0)        # the direct kernels unless a test asks for Winograd (algo=2)
    pc = _PackedConv(torch.device(DEV), w, b, ksize, stride, splits, shuffle=shuffle)
    plan = DenoiserPlan.__new__(DenoiserPlan)
    from yond_public_amd import _lib as L
    plan.lib = L.load()
    plan.dev = torch.device(DEV)
    if shuffle:
        dst = torch.full((N, 2 * H, 2 * W, pc.cout_real_p), float('nan'), device=DEV)
    elif stride == 2:
        dst = torch.full((N, (H + 1) // 2, (W + 1) // 2, pc.coutp), float('nan'), device=DEV)
    else:
        dst = torch.full((N, H, W, pc.coutp), float('nan'), device=DEV)
    plan._conv(pc, xs[0], xs[1] if len(xs) > 1 else None, N, H, W, dst, **kw)
    torch.cuda.synchronize()
    return dst.cpu()


@pytest.mark.parametrize("C,Co,N,H,W", [(32, 32, 1, 24, 40), (64, 64, 2, 16, 32), (32, 64, 1, 9, 33), (128, 128, 1, 8, 32)])
def test_conv3x3_s1_plain(C, Co, N, H, W):
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    got = nchw(run_conv(w, b, 3, 1, [C], [nhwc(x).to(DEV)], N, H, W))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert report(f"conv3x3 s1 C{C}->{Co} {H}x{W}", got, ref) < 2e-5


def test_conv3x3_s1_fused_film_silu_residual():
    g = torch.Generator().manual_seed(7)
    N, C, H, W = 2, 64, 16, 48
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    es, et = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    res = torch.randn(N, C, H, W, generator=g)
    got = nchw(run_conv(w, None, 3, 1, [C], [nhwc(x).to(DEV)], N, H, W, escale=es.to(DEV), eshift=et.to(DEV), ebatch=1,
                        res=nhwc(res).to(DEV), pre_act=1, post_act=2, slope=0.3))
    z = F.conv2d(F.silu(x.double()), w.double(), padding=1)
    z = F.leaky_relu(z * es.double()[:, :, None, None] + et.double()[:, :, None, None], 0.3) + res.double()
    assert report("conv3x3 fused", got, z) < 2e-5


def test_conv3x3_two_source_leaky():
    g = torch.Generator().manual_seed(9)
    N, C, H, W = 1, 32, 16, 32
    a, b2 = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 2 * C, 3, 3, generator=g) / (3 * (2 * C) ** 0.5)
    b = torch.randn(C, generator=g)
    got = nchw(run_conv(w, b, 3, 1, [C, C], [nhwc(a).to(DEV), nhwc(b2).to(DEV)], N, H, W, post_act=2, slope=0.2))
    ref = F.leaky_relu(F.conv2d(torch.cat([a, b2], 1).double(), w.double(), b.double(), padding=1), 0.2)
    assert report("conv3x3 two-source", got, ref) < 2e-5


@pytest.mark.parametrize("C,Co,N,H,W", [(64, 64, 1, 8, 32), (64, 64, 2, 16, 64), (32, 64, 1, 9, 33), (128, 128, 1, 24, 40),
                                        (256, 64, 1, 94, 126), (32, 32, 2, 24, 72), (64, 32, 1, 17, 40),
                                        (64, 64, 1, 1, 1), (64, 64, 1, 2, 3), (128, 64, 2, 4, 4),
                                        (32, 32, 1, 200, 352), (64, 64, 1, 136, 288)])      # > 256 tiles: several tiles per workgroup
def test_conv3x3_winograd_plain(C, Co, N, H, W):
    """Winograd F(2x2,3x3) kernel (algo 1) against the float64 direct convolution: same tolerance as the direct
    kernel; partial tiles, odd extents, several tiles per workgroup."""
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    got = nchw(run_conv(w, b, 3, 1, [C], [nhwc(x).to(DEV)], N, H, W, algo=2))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert report(f"winograd conv3x3 C{C}->{Co} {H}x{W}", got, ref) < 2e-5


def test_conv3x3_winograd_fused_two_source():
    g = torch.Generator().manual_seed(17)
    N, C, H, W = 2, 64, 20, 72
    a, b2 = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 2 * C, 3, 3, generator=g) / (3 * (2 * C) ** 0.5)
    es, et = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    res = torch.randn(N, C, H, W, generator=g)
    got = nchw(run_conv(w, None, 3, 1, [C, C], [nhwc(a).to(DEV), nhwc(b2).to(DEV)], N, H, W, escale=es.to(DEV), eshift=et.to(DEV),
                        ebatch=1, res=nhwc(res).to(DEV), pre_act=1, post_act=2, slope=0.3, algo=2))
    z = F.conv2d(F.silu(torch.cat([a, b2], 1).double()), w.double(), padding=1)
    z = F.leaky_relu(z * es.double()[:, :, None, None] + et.double()[:, :, None, None], 0.3) + res.double()
    assert report("winograd conv3x3 fused two-source", got, z) < 2e-5


@pytest.mark.parametrize("C,Co,N,H,W", [(64, 64, 1, 8, 32), (64, 64, 2, 16, 64), (32, 64, 1, 9, 33), (128, 128, 1, 24, 40),
                                        (256, 64, 1, 94, 126), (32, 32, 2, 24, 72), (64, 32, 1, 17, 40), (32, 32, 1, 5, 7),
                                        (64, 64, 1, 1, 1), (64, 64, 1, 2, 3), (128, 64, 2, 4, 4),
                                        (32, 32, 1, 200, 352), (64, 64, 1, 136, 288),       # > 256 tiles: several tiles per workgroup
                                        (64, 64, 1, 190, 520), (128, 128, 2, 94, 126)])     # >= 256 12-row tiles: the three-rows-per-wave shape
def test_conv3x3_split_plain(C, Co, N, H, W):
    """Split-operand fp16-MFMA kernel (algo 3: a = h + l 2^-11, three fp16 products per fp32 product, fp32 accumulate)
    against the float64 convolution: the SAME tolerance as the fp32 kernels."""
    g = torch.Generator().manual_seed(C + H + 3)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    got = nchw(run_conv(w, b, 3, 1, [C], [nhwc(x).to(DEV)], N, H, W, algo='split'))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert report(f"split conv3x3 C{C}->{Co} {N}x{H}x{W}", got, ref) < 2e-5


@pytest.mark.parametrize("C,Co,N,H,W", [(64, 64, 70, 16, 16), (128, 128, 9, 8, 8), (64, 128, 5, 13, 11), (256, 256, 3, 6, 16)])
def test_conv3x3_split_folded_tiles_nhwc(C, Co, N, H, W):
    """The plain [N][H][W][C] 3x3 layer on a batch of images at most 16 pixels wide (training patches' deep levels): folded tiles -- two or
    four sub-tiles, of one image or of several, share the MFMA's 32-pixel row.  Bias + LeakyReLU + residual against float64, and every image
    of the batch BIT-EQUAL to the same image convolved alone (another grouping of the sub-tiles, the same arithmetic per pixel)."""
    g = torch.Generator().manual_seed(C + H + N)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    res = torch.randn(N, Co, H, W, generator=g)
    xd, rd = nhwc(x).to(DEV), nhwc(res).to(DEV)
    got = run_conv(w, b, 3, 1, [C], [xd], N, H, W, res=rd, post_act=2, slope=0.2, algo='split')
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.2) + res.double()
    assert report(f"folded split conv3x3 C{C}->{Co} {N}x{H}x{W}", nchw(got), ref) < 2e-5
    for n in (0, N // 2, N - 1):
        one = run_conv(w, b, 3, 1, [C], [xd[n:n + 1].contiguous()], 1, H, W, res=rd[n:n + 1].contiguous(), post_act=2, slope=0.2, algo='split')
        assert torch.equal(one[0], got[n]), n


def test_conv3x3_split_accuracy_beside_fp32_kernels():
    """Error against float64 of the three 3x3 kernels on the same data, at unit scale and at magnitudes where the
    fp16 halves are subnormal (1e-6) or large (1e3): the split kernel must be no worse than 2x the fp32 direct
    kernel's error or no worse than the fp32 Winograd kernel's (whichever is larger); where whole operands lie below
    fp16's normal range the documented absolute floor applies instead."""
    g = torch.Generator().manual_seed(11)
    N, C, H, W = 1, 128, 32, 64
    x0 = torch.randn(N, C, H, W, generator=g)
    w0 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    for sx, sw in ((1.0, 1.0), (1e-6, 1.0), (1e3, 1.0), (1.0, 1e-4), (3e-5, 30.0)):
        x, w = x0 * sx, w0 * sw
        ref = F.conv2d(x.double(), w.double(), padding=1)
        scale = ref.abs().max().item()
        err = {}
        for name, algo in (('direct', 0), ('winograd', 2), ('split', 'split')):
            got = nchw(run_conv(w, None, 3, 1, [C], [nhwc(x).to(DEV)], N, H, W, algo=algo)).double()
            err[name] = ((got - ref).abs().max().item() / scale, ((got - ref) ** 2).mean().sqrt().item() / scale)
        print(f"[accuracy] x*{sx:g} w*{sw:g}: max / rms error relative to max|ref|: " +
              ", ".join(f"{k} {v[0]:.2e} / {v[1]:.2e}" for k, v in err.items()))
        # an operand below fp16's normal range (|a| < 6.1e-5) keeps an ABSOLUTE error of 2^-36 instead of a relative one:
        # per output that is 2^-36 * sqrt(9 C) * rms(other operand), x6 for the maximum over the tensor
        w_rms = sw / (3 * C ** 0.5)
        floor = 2.0 ** -36 * 6 * (9 * C) ** 0.5 * ((w_rms if sx < 1e-3 else 0.0) + (sx if w_rms < 1e-3 else 0.0)) / scale
        assert err['split'][0] <= max(2.0 * err['direct'][0], err['winograd'][0], floor), (err, floor)
        assert err['split'][1] <= max(2.0 * err['direct'][1], err['winograd'][1], floor), (err, floor)


def test_conv3x3_split_fused_two_source():
    g = torch.Generator().manual_seed(13)
    N, C, H, W = 2, 64, 20, 48
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    es, et = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    res = torch.randn(N, C, H, W, generator=g)
    got = nchw(run_conv(w, None, 3, 1, [C], [nhwc(x).to(DEV)], N, H, W, escale=es.to(DEV), eshift=et.to(DEV), ebatch=1,
                        res=nhwc(res).to(DEV), pre_act=1, post_act=2, slope=0.3, algo='split'))
    z = F.conv2d(F.silu(x.double()), w.double(), padding=1)
    z = F.leaky_relu(z * es.double()[:, :, None, None] + et.double()[:, :, None, None], 0.3) + res.double()
    assert report("split conv3x3 fused", got, z) < 2e-5
    a, b2 = torch.randn(1, 32, 16, 32, generator=g), torch.randn(1, 32, 16, 32, generator=g)
    w2 = torch.randn(32, 64, 3, 3, generator=g) / (3 * 64 ** 0.5)
    b = torch.randn(32, generator=g)
    got = nchw(run_conv(w2, b, 3, 1, [32, 32], [nhwc(a).to(DEV), nhwc(b2).to(DEV)], 1, 16, 32, post_act=2, slope=0.2, algo='split'))
    ref = F.leaky_relu(F.conv2d(torch.cat([a, b2], 1).double(), w2.double(), b.double(), padding=1), 0.2)
    assert report("split conv3x3 two-source", got, ref) < 2e-5


@pytest.mark.parametrize("C,H,W", [(32, 40, 70), (64, 50, 96), (128, 24, 64)])
def test_conv3x3_split_silu_epilogue(C, H, W):
    """Descriptor post_act 1: SiLU of the stored value (the inner activation of a residual block in conv1's epilogue).
    conv(post_act 1) followed by conv(pre_act 0) must equal conv(post_act 0) followed by conv(pre_act 1) bit for bit --
    the same fp32 function of the same value -- and match the float64 block; other kernels refuse the flag."""
    g = torch.Generator().manual_seed(C + W)
    N = 2
    x = torch.randn(N, C, H, W, generator=g)
    w1 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    w2 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    es, et = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    xd = nhwc(x).to(DEV)
    kw = dict(escale=es.to(DEV), eshift=et.to(DEV), ebatch=1, pre_act=1, algo='split')
    t_act = run_conv(w1, None, 3, 1, [C], [xd], N, H, W, post_act=1, **kw)
    t_raw = run_conv(w1, None, 3, 1, [C], [xd], N, H, W, **kw)
    out_a = run_conv(w2, None, 3, 1, [C], [t_act.to(DEV)], N, H, W, res=xd, algo="split")
    out_b = run_conv(w2, None, 3, 1, [C], [t_raw.to(DEV)], N, H, W, res=xd, pre_act=1, algo="split")
    assert torch.equal(out_a, out_b)
    z = F.conv2d(F.silu(x.double()), w1.double(), padding=1) * es.double()[:, :, None, None] + et.double()[:, :, None, None]
    assert report(f"split conv3x3 SiLU epilogue C{C}", nchw(t_act), F.silu(z)) < 2e-5
    ref = F.conv2d(F.silu(z), w2.double(), padding=1) + x.double()
    assert report(f"residual block with the SiLU in the producer C{C}", nchw(out_a), ref) < 4e-5
    with pytest.raises(Exception):
        run_conv(w1, None, 3, 1, [C], [xd], N, H, W, post_act=1, algo=0)


def sp_decode(sp, N, C, H, W):
    """A split-plane tensor (YondConvDesc out_fmt 1) back to [N][C][H][W] float64: h + l * 2^-11; also returns the pad units."""
    from yond_public_amd.engine import sp_plane_units
    ps = sp_plane_units(H, W)
    u = sp.cpu().view(torch.float16).reshape(N, C // 16, 2, 2, ps, 8)
    v = u[..., :H * W, :].double()
    val = v[:, :, :, 0] + v[:, :, :, 1] / 2048.0                      # [N][C/16][half][H*W][8]
    val = val.permute(0, 1, 2, 4, 3).reshape(N, C, H, W)               # channel = c16*16 + half*8 + j
    return val, u[..., H * W:, :]


@pytest.mark.parametrize("C,N,H,W,out4", [(64, 2, 40, 70, False), (128, 1, 380, 100, False), (32, 2, 50, 75, False), (32, 1, 37, 70, True),
                                          (256, 1, 30, 61, False), (64, 5, 16, 16, False), (128, 3, 8, 8, False), (64, 3, 13, 11, False), (64, 7, 6, 5, False)])
def test_conv3x3_split_planes_pair(C, N, H, W, out4):
    """The block-internal tensor of a residual block in SPLIT PLANES: conv1 stores SiLU(FiLM(conv1)) as the (h, l) halves its
    consumer would stage (out_fmt 1), conv2 stages them by LDS-DMA alone (in_fmt 1).  The pair must equal the float32-NHWC
    pair BIT FOR BIT (the same split of the same float32 value), the stored planes must decode to the NHWC tensor to 2^-22,
    the zero pads must stay zero; every tile shape of the kernel (8 / 12 / 16-row tiles, partial tiles, batch), with and without
    the fused output projection."""
    from yond_public_amd.engine import _PackedConv, DenoiserPlan, sp_plane_units
    from yond_public_amd import _lib as L
    g = torch.Generator().manual_seed(C + W)
    x = torch.randn(N, C, H, W, generator=g)
    w1 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    w2 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    es, et = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    es2, et2 = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    xd = nhwc(x).to(DEV)
    plan = DenoiserPlan.__new__(DenoiserPlan)
    plan.lib, plan.dev = L.load(), torch.device(DEV)
    pc1 = _PackedConv(plan.dev, w1, None, 3, 1, [C])
    pc2 = _PackedConv(plan.dev, w2, None, 3, 1, [C])
    kw1 = dict(escale=es.to(DEV), eshift=et.to(DEV), ebatch=1, pre_act=1, post_act=1, algo='split')
    kw2 = dict(escale=es2.to(DEV), eshift=et2.to(DEV), ebatch=1, res=xd, algo='split')
    t_nhwc = torch.full((N, H, W, C), float('nan'), device=DEV)
    plan._conv(pc1, xd, None, N, H, W, t_nhwc, **kw1)
    t_sp = plan._new_sp('t', N, H, W, C)
    plan._conv(pc1, xd, None, N, H, W, t_sp, out_fmt=1, **kw1)
    if out4:
        w4 = (torch.randn(4, C, generator=g) / C ** 0.5).to(DEV)
        b4 = torch.randn(4, generator=g).to(DEV)
        xin = torch.rand(N, H, W, 4, generator=g).to(DEV)
        ub = (torch.rand(N, generator=g) + 0.5).to(DEV)
        o_a = torch.full((N, H, W, 4), float('nan'), device=DEV)
        o_b = torch.full((N, H, W, 4), float('nan'), device=DEV)
        plan._conv(pc2, t_nhwc, None, N, H, W, None, out4=(w4, b4, xin, ub, o_a), **kw2)
        plan._conv(pc2, t_sp, None, N, H, W, None, out4=(w4, b4, xin, ub, o_b), in_fmt=1, **kw2)
    else:
        o_a = torch.full((N, H, W, C), float('nan'), device=DEV)
        o_b = torch.full((N, H, W, C), float('nan'), device=DEV)
        plan._conv(pc2, t_nhwc, None, N, H, W, o_a, **kw2)
        plan._conv(pc2, t_sp, None, N, H, W, o_b, in_fmt=1, **kw2)
    torch.cuda.synchronize()
    val, pads = sp_decode(t_sp, N, C, H, W)
    assert not pads.view(torch.int16).any()
    ref_t = nchw(t_nhwc.cpu()).double()
    assert float((val - ref_t).abs().max()) <= 2.0 ** -21 * max(1.0, float(ref_t.abs().max()))
    assert torch.equal(o_a.cpu(), o_b.cpu()), float((o_a - o_b).abs().max())
    if not out4:
        z = F.conv2d(F.silu(x.double()), w1.double(), padding=1) * es.double()[:, :, None, None] + et.double()[:, :, None, None]
        ref = F.conv2d(F.silu(z), w2.double(), padding=1) * es2.double()[:, :, None, None] + et2.double()[:, :, None, None] + x.double()
        assert report(f"residual block through split planes C{C} {H}x{W}", nchw(o_b.cpu()), ref) < 6e-5
    # refused where it is not built: other kernels, a pre-activation on split-plane input, a residual on split-plane output
    with pytest.raises(Exception):
        plan._conv(pc2, t_sp, None, N, H, W, o_b if not out4 else torch.empty(N, H, W, C, device=DEV), in_fmt=1, algo=0)
    with pytest.raises(Exception):
        plan._conv(pc2, t_sp, None, N, H, W, torch.empty(N, H, W, C, device=DEV), in_fmt=1, pre_act=1, algo='split')
    with pytest.raises(Exception):
        plan._conv(pc1, xd, None, N, H, W, t_sp, out_fmt=1, res=xd, algo='split')


def to_sp(x):
    """[N][H][W][C] float32 (CPU) -> split planes (float32-typed buffer on the device), the exact halves the staging forms."""
    from yond_public_amd.engine import sp_plane_units
    N, H, W, C = x.shape
    ps = sp_plane_units(H, W)
    h = x.half()
    l = ((x - h.float()) * 2048.0).half()
    out = torch.zeros(N, C // 16, 2, 2, ps, 8, dtype=torch.float16)
    for part, t in enumerate((h, l)):
        out[:, :, :, part, :H * W, :] = t.reshape(N, H * W, C // 16, 2, 8).permute(0, 2, 3, 1, 4)
    return out.reshape(-1).view(torch.float32).to(DEV)


def sp_halves(sp, N, C, H, W):
    """split planes -> (h, l) as [N][H][W][C] float16 tensors (CPU)."""
    from yond_public_amd.engine import sp_plane_units
    u = sp.cpu().view(torch.float16).reshape(N, C // 16, 2, 2, sp_plane_units(H, W), 8)[:, :, :, :, :H * W, :]
    return [u[:, :, :, part].permute(0, 3, 1, 2, 4).reshape(N, H, W, C) for part in (0, 1)]


def to_p4(x):
    """[N][H][W][C] -> planes of 4 channels [N][C/4][H*W][4] (device)."""
    N, H, W, C = x.shape
    return x.reshape(N, H * W, C // 4, 4).permute(0, 2, 1, 3).contiguous().to(DEV)


def from_p4(p, N, H, W, C):
    return p.cpu().reshape(N, C // 4, H * W, 4).permute(0, 2, 1, 3).reshape(N, H, W, C)


def bare_plan():
    from yond_public_amd.engine import DenoiserPlan
    from yond_public_amd import _lib as L
    plan = DenoiserPlan.__new__(DenoiserPlan)
    plan.lib, plan.dev = L.load(), torch.device(DEV)
    return plan


@pytest.mark.parametrize("C,N,H,W", [(64, 2, 40, 70), (128, 1, 380, 100), (32, 2, 50, 75),
                                     # images at most 16 pixels wide: folded tiles (two 16-column sub-tiles per MFMA row, of one image or of two)
                                     (64, 5, 16, 16), (128, 3, 8, 8), (256, 8, 16, 16), (64, 3, 24, 16), (64, 1, 13, 11), (128, 1, 8, 16), (512, 7, 8, 8), (64, 2, 5, 6)])
def test_conv3x3_split_plane_flow_block(C, N, H, W):
    """The residual block in the formats of the default data flow: x in planes of 4 channels (conv1's staged input, conv2's
    residual), tmp and out in split planes.  Every tensor must hold exactly what the [N][H][W][C] float32 path computes: tmp and
    out decode to the split of the float32 results, bit for bit."""
    from yond_public_amd.engine import _PackedConv
    g = torch.Generator().manual_seed(C + H)
    x = nhwc(torch.randn(N, C, H, W, generator=g))
    w1 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    w2 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    es, et = torch.randn(N, C, generator=g).to(DEV), torch.randn(N, C, generator=g).to(DEV)
    es2, et2 = torch.randn(N, C, generator=g).to(DEV), torch.randn(N, C, generator=g).to(DEV)
    plan = bare_plan()
    pc1, pc2 = _PackedConv(plan.dev, w1, None, 3, 1, [C]), _PackedConv(plan.dev, w2, None, 3, 1, [C])
    xd, xp = x.to(DEV), to_p4(x)
    kw1 = dict(escale=es, eshift=et, ebatch=1, pre_act=1, post_act=1, algo='split')
    kw2 = dict(escale=es2, eshift=et2, ebatch=1, algo='split')
    t_ref = torch.empty(N, H, W, C, device=DEV)
    o_ref = torch.empty(N, H, W, C, device=DEV)
    plan._conv(pc1, xd, None, N, H, W, t_ref, **kw1)
    plan._conv(pc2, t_ref, None, N, H, W, o_ref, res=xd, **kw2)
    t_sp, o_sp = plan._new_sp('t', N, H, W, C), plan._new_sp('o', N, H, W, C)
    plan._conv(pc1, xp, None, N, H, W, t_sp, in_fmt=2, out_fmt=1, **kw1)
    plan._conv(pc2, t_sp, None, N, H, W, o_sp, res=xp, in_fmt=1, out_fmt=1, res_fmt=2, **kw2)
    torch.cuda.synchronize()
    for name, sp, ref in (("tmp", t_sp, t_ref), ("out", o_sp, o_ref)):
        h, l = sp_halves(sp, N, C, H, W)
        rh = ref.cpu().half()
        rl = ((ref.cpu() - rh.float()) * 2048.0).half()
        assert torch.equal(h.view(torch.int16), rh.view(torch.int16)), name
        assert torch.equal(l.view(torch.int16), rl.view(torch.int16)), name
    with pytest.raises(Exception):          # a split-plane store takes its residual in planes of 4 only
        plan._conv(pc2, t_sp, None, N, H, W, o_sp, res=xd, in_fmt=1, out_fmt=1, **kw2)


@pytest.mark.parametrize("C,N,H,W", [(32, 2, 37, 70), (64, 1, 64, 130), (128, 1, 23, 45),
                                     # batches of small images (output at most 16 pixels wide, more than 256 plain tiles): folded tiles, 2 / 4 sub-tiles per MFMA row
                                     (64, 40, 32, 32), (128, 40, 16, 16), (64, 45, 30, 27), (64, 70, 11, 13)])
def test_conv3x3_s2_split_plane_input_planes4_output(C, N, H, W):
    """Stride-2 layer of the default data flow: input in split planes (raw block output) staged by LDS-DMA, output in planes of
    4 channels -- equal to the [N][H][W][C] path bit for bit, in both output formats; the second output (SiLU of the value in split
    planes, YondConvDesc.dst2) decodes to SiLU of the first."""
    from yond_public_amd.engine import _PackedConv
    g = torch.Generator().manual_seed(C + W)
    x = nhwc(torch.randn(N, C, H, W, generator=g))
    w = torch.randn(2 * C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(2 * C, generator=g)
    plan = bare_plan()
    pc = _PackedConv(plan.dev, w, b, 3, 2, [C])
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    ref = torch.empty(N, Ho, Wo, 2 * C, device=DEV)
    plan._conv(pc, x.to(DEV), None, N, H, W, ref, algo='split')
    got = torch.empty(N, Ho, Wo, 2 * C, device=DEV)
    plan._conv(pc, to_sp(x), None, N, H, W, got, algo='split', in_fmt=1)
    gp4 = torch.empty(N * 2 * C * Ho * Wo, device=DEV)
    plan._conv(pc, to_sp(x), None, N, H, W, gp4, algo='split', in_fmt=1, out_fmt=2)
    gp4b = torch.empty(N * 2 * C * Ho * Wo, device=DEV)
    second = plan._new_sp('second', N, Ho, Wo, 2 * C)
    plan._conv(pc, to_sp(x), None, N, H, W, gp4b, algo='split', in_fmt=1, out_fmt=2, dst2=second)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ref.cpu())
    assert torch.equal(from_p4(gp4, N, Ho, Wo, 2 * C), ref.cpu())
    assert torch.equal(gp4b.cpu(), gp4.cpu())
    val, pads = sp_decode(second, N, 2 * C, Ho, Wo)
    assert not pads.view(torch.int16).any()
    want = F.silu(nchw(ref.cpu()).double())
    assert float((val - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
    z = F.conv2d(nchw(x).double(), w.double(), b.double(), stride=2, padding=1)
    assert report(f"stride-2 from split planes C{C}", nchw(got.cpu()), z) < 2e-5


@pytest.mark.parametrize("c,h,w,N", [(64, 24, 40, 2), (32, 19, 33, 2), (128, 9, 35, 2),
                                     # batches of small images (at most 16 pixels wide, more than 256 plain tiles): folded tiles, 2 / 4 sub-tiles per MFMA row
                                     (64, 16, 16, 40), (128, 8, 8, 40), (64, 13, 11, 45), (64, 5, 7, 70)])
def test_decoder_gemm_split_plane_inputs(c, h, w, N):
    """The decoder GEMM (ConvTranspose2d 2x2 + cat + 1x1 shortcut folded) with BOTH sources in split planes -- the
    low-resolution tensor and the skip tensor gathered at the sub-position -- and the output in either format."""
    from yond_public_amd.engine import _PackedConv
    g = torch.Generator().manual_seed(c + h)
    cur = nhwc(torch.randn(N, 2 * c, h, w, generator=g))
    skip = nhwc(torch.randn(N, c, 2 * h, 2 * w, generator=g))
    wf = torch.randn(3 * c, c, 2, 2, generator=g) / (3 * c) ** 0.5        # ConvTranspose2d layout over [cur | skip]
    bf = torch.randn(c, generator=g)
    plan = bare_plan()
    pc = _PackedConv(plan.dev, wf, bf, 1, 1, [2 * c, c], shuffle=True)
    ref = torch.empty(N, 2 * h, 2 * w, c, device=DEV)
    plan._conv(pc, cur.to(DEV), skip.to(DEV), N, h, w, ref, algo='split')
    got = torch.empty(N, 2 * h, 2 * w, c, device=DEV)
    plan._conv(pc, to_sp(cur), to_sp(skip), N, h, w, got, algo='split', in_fmt=1)
    gp4 = torch.empty(N * c * 4 * h * w, device=DEV)
    plan._conv(pc, to_sp(cur), to_sp(skip), N, h, w, gp4, algo='split', in_fmt=1, out_fmt=2)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ref.cpu())
    assert torch.equal(from_p4(gp4, N, 2 * h, 2 * w, c), ref.cpu())
    if c >= 64 and w > 16:
        # the second output (YondConvDesc.dst2): SiLU of the value in split planes, what the next block's conv1 stages by LDS-DMA
        gp4b = torch.empty(N * c * 4 * h * w, device=DEV)
        second = plan._new_sp('second', N, 2 * h, 2 * w, c)
        plan._conv(pc, to_sp(cur), to_sp(skip), N, h, w, gp4b, algo='split', in_fmt=1, out_fmt=2, dst2=second)
        torch.cuda.synchronize()
        assert torch.equal(gp4b.cpu(), gp4.cpu())
        val, pads = sp_decode(second, N, c, 2 * h, 2 * w)
        assert not pads.view(torch.int16).any()
        want = F.silu(nchw(ref.cpu()).double())
        assert float((val - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


def test_conv3x3_split_fused_12_row_tiles():
    """FiLM + SiLU + LeakyReLU + residual on a layer large enough for the 12 x 32-pixel tile shape (partial tiles both ways)."""
    g = torch.Generator().manual_seed(19)
    N, C, H, W = 1, 64, 188, 540
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    es, et = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    res = torch.randn(N, C, H, W, generator=g)
    got = nchw(run_conv(w, None, 3, 1, [C], [nhwc(x).to(DEV)], N, H, W, escale=es.to(DEV), eshift=et.to(DEV), ebatch=1,
                        res=nhwc(res).to(DEV), pre_act=1, post_act=2, slope=0.3, algo='split'))
    z = F.conv2d(F.silu(x.double()), w.double(), padding=1)
    z = F.leaky_relu(z * es.double()[:, :, None, None] + et.double()[:, :, None, None], 0.3) + res.double()
    assert report("split conv3x3 fused, 12-row tiles", got, z) < 2e-5


@pytest.mark.parametrize("with_x,with_ub,res", [(True, True, True), (False, False, False), (True, False, True)])
def test_conv3x3_split_fused_output_projection(with_x, with_ub, res):
    """The 1x1 output projection (conv10 + global residual + de-normalisation) computed in the epilogue of the last 3x3
    convolution must be BIT-identical to the convolution followed by yond_conv_out_f32."""
    from yond_public_amd import _lib as L
    from yond_public_amd.engine import _PackedConv, DenoiserPlan
    lib = L.load()
    g = torch.Generator().manual_seed(23)
    N, C, H, W = 2, 32, 37, 70
    x = nhwc(torch.randn(N, C, H, W, generator=g)).to(DEV)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(C, generator=g)
    r = nhwc(torch.randn(N, C, H, W, generator=g)).to(DEV) if res else None
    w4 = (torch.randn(4, C, generator=g) / C ** 0.5).to(DEV)
    b4 = torch.randn(4, generator=g).to(DEV)
    xin = torch.rand(N, H, W, 4, generator=g).to(DEV) if with_x else None
    ub = (torch.rand(N, generator=g) + 0.5).to(DEV) if with_ub else None
    pc = _PackedConv(torch.device(DEV), w, b, 3, 1, [C])
    plan = DenoiserPlan.__new__(DenoiserPlan)
    plan.lib, plan.dev = lib, torch.device(DEV)
    feat = torch.empty(N, H, W, C, device=DEV)
    plan._conv(pc, x, None, N, H, W, feat, res=r, pre_act=1, post_act=2, slope=0.2, algo='split')
    ref = torch.full((N, H, W, 4), float('nan'), device=DEV)
    L.check(lib.yond_conv_out_f32(L.ptr(feat), C, L.ptr(w4), L.ptr(b4), L.ptr(xin), L.ptr(ub), N, H, W, L.ptr(ref), L.stream()), "conv_out")
    got = torch.full((N, H, W, 4), float('nan'), device=DEV)
    plan._conv(pc, x, None, N, H, W, None, res=r, pre_act=1, post_act=2, slope=0.2, algo='split', out4=(w4, b4, xin, ub, got))
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ref.cpu()), float((got - ref).abs().max())


def test_conv3x3_half_staged_path():
    """algo 4: the same kernel with the h halves only = plain fp16 MFMA (cfg 5), fp16-level tolerance."""
    g = torch.Generator().manual_seed(17)
    N, C, H, W = 1, 64, 24, 40
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    got = nchw(run_conv(w, None, 3, 1, [C], [nhwc(x).to(DEV)], N, H, W, algo='half'))
    ref = F.conv2d(x.double(), w.double(), padding=1)
    assert report("half-staged conv3x3", got, ref) < 5e-3


@pytest.mark.parametrize("C,Co,N,H,W", [(128, 128, 1, 400, 230), (64, 256, 2, 37, 70), (256, 128, 1, 24, 64), (128, 384, 1, 13, 33)])
def test_conv3x3_half_128_channel_tiles(C, Co, N, H, W):
    """algo 4 with 128-channel tiles (h-only operands leave one accumulator per block: a wave owns TWO 32-channel blocks; the fp16 path's
    layers with >= 128 output channels, engine.HALF_TN128): both tile heights (12-row tiles from 256 tiles on, else 8-row), SiLU staging,
    FiLM + residual epilogue, ragged edges, batch -- equal to the 64-channel-tile form of the same arithmetic to accumulation order, and
    at fp16 level from float64."""
    from yond_public_amd import engine as E
    g = torch.Generator().manual_seed(C + Co + W)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    es, et = torch.randn(N, Co, generator=g).to(DEV), torch.randn(N, Co, generator=g).to(DEV)
    r = torch.randn(N, H, W, Co, generator=g).to(DEV)
    xd = nhwc(x).to(DEV)
    out = {}
    for flag in (True, False):
        E.HALF_TN128 = flag
        try:
            out[flag] = (run_conv(w, b, 3, 1, [C], [xd], N, H, W, algo='half'),
                         run_conv(w, None, 3, 1, [C], [xd], N, H, W, algo='half', pre_act=1, escale=es, eshift=et, ebatch=1, res=r))
        finally:
            E.HALF_TN128 = True
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref2 = F.conv2d(F.silu(x.double()), w.double(), padding=1) * es.cpu().double()[:, :, None, None] + et.cpu().double()[:, :, None, None] + nchw(r.cpu()).double()
    assert report(f"half conv3x3, 128-channel tiles C{C}->{Co} {H}x{W} vs 64-channel tiles", out[True][0], out[False][0]) <= 2e-5 * float(ref.abs().max())
    assert report(f"half conv3x3, 128-channel tiles, SiLU + FiLM + residual vs 64-channel tiles", out[True][1], out[False][1]) <= 2e-5 * float(ref2.abs().max())
    assert report(f"half conv3x3, 128-channel tiles vs float64", nchw(out[True][0]), ref) < 5e-3 * max(1.0, float(ref.abs().max()))
    assert report(f"half conv3x3, 128-channel tiles, fused vs float64", nchw(out[True][1]), ref2) < 5e-3 * max(1.0, float(ref2.abs().max()))


@pytest.mark.parametrize("ks,st,C,Co", [(3, 1, 64, 64), (3, 1, 32, 32), (3, 2, 32, 64), (1, 1, 64, 32)])
def test_conv_fp16_mfma_path(ks, st, C, Co):
    """algo 'fp16' (descriptor algo 2, BASELINE cfg 5): operands rounded to half at the matrix core, fp32 accumulate.
    Against the float64 convolution of the half-rounded operands the error is accumulation order only; against the
    unrounded one it is the half rounding (relative 2^-11 per operand)."""
    g = torch.Generator().manual_seed(ks * 10 + st + C)
    N, H, W = 1, 24, 40
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, ks, ks, generator=g) / (ks * C ** 0.5)
    b = torch.randn(Co, generator=g)
    got = nchw(run_conv(w, b, ks, st, [C], [nhwc(x).to(DEV)], N, H, W, algo='fp16'))
    pad = ks // 2
    xh, wh = x.half().double(), w.half().double()
    ref_h = F.conv2d(xh, wh, b.double(), stride=st, padding=pad)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=pad)
    assert report(f"fp16-mfma conv {ks}x{ks} s{st} C{C}->{Co} vs half-rounded operands", got, ref_h) < 2e-5
    assert report(f"fp16-mfma conv {ks}x{ks} s{st} C{C}->{Co} vs fp32 operands", got, ref) < 5e-3


@pytest.mark.parametrize("C,N,H,W", [(32, 1, 16, 64), (64, 2, 32, 32), (32, 1, 18, 34)])
def test_conv3x3_s2(C, N, H, W):
    g = torch.Generator().manual_seed(C + W)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(2 * C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(2 * C, generator=g)
    got = nchw(run_conv(w, b, 3, 2, [C], [nhwc(x).to(DEV)], N, H, W))
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    assert report(f"conv3x3 s2 C{C} {H}x{W}", got, ref) < 2e-5


@pytest.mark.parametrize("C,N,H,W", [(32, 1, 16, 64), (64, 2, 32, 32), (32, 1, 18, 34), (128, 1, 7, 9), (32, 1, 2, 2),
                                     (32, 1, 136, 544), (64, 1, 33, 65),       # odd sizes; > 256 tiles
                                     (64, 40, 32, 32), (128, 40, 16, 16), (64, 70, 11, 13)])    # batches of small images: folded tiles
def test_conv3x3_s2_split(C, N, H, W):
    """Stride-2 form of the split-operand kernel (algo 3), same tolerance as the fp32 kernel; an image of a batch is bit-equal to the
    image convolved alone (folded tiles group sub-tiles of several images into one MFMA row)."""
    g = torch.Generator().manual_seed(C + W + 1)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(2 * C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(2 * C, generator=g)
    xd = nhwc(x).to(DEV)
    got = run_conv(w, b, 3, 2, [C], [xd], N, H, W, algo='split')
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    assert report(f"split conv3x3 s2 C{C} {N}x{H}x{W}", nchw(got), ref) < 2e-5
    if N > 2:
        for n in (0, N - 1):
            one = run_conv(w, b, 3, 2, [C], [xd[n:n + 1].contiguous()], 1, H, W, algo='split')
            assert torch.equal(one[0], got[n]), n


@pytest.mark.parametrize("algo", [0, 'split'])
@pytest.mark.parametrize("C", [32, 128])
def test_conv1x1_two_source(C, algo):
    g = torch.Generator().manual_seed(C)
    N, H, W = 2, 8, 40
    a, b2 = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 2 * C, 1, 1, generator=g) / (2 * C) ** 0.5
    b = torch.randn(C, generator=g)
    got = nchw(run_conv(w, b, 1, 1, [C, C], [nhwc(a).to(DEV), nhwc(b2).to(DEV)], N, H, W, algo=algo))
    ref = F.conv2d(torch.cat([a, b2], 1).double(), w.double(), b.double())
    assert report(f"conv1x1 two-source C{C} algo {algo}", got, ref) < 2e-5


@pytest.mark.parametrize("algo", [0, 'split'])
@pytest.mark.parametrize("Ci,Co", [(64, 32), (256, 128)])
def test_conv_transpose_2x2(Ci, Co, algo):
    g = torch.Generator().manual_seed(Ci)
    N, H, W = 2, 8, 24
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5
    b = torch.randn(Co, generator=g)
    got = nchw(run_conv(w, b, 1, 1, [Ci], [nhwc(x).to(DEV)], N, H, W, shuffle=True, algo=algo))
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    assert report(f"convT {Ci}->{Co} algo {algo}", got, ref) < 2e-5


@pytest.mark.parametrize("c,h,w", [(32, 9, 20), (64, 9, 20), (128, 11, 37), (64, 40, 130)])
@pytest.mark.parametrize("algo", [0, 'split'])
def test_conv_transpose_fused_with_skip_shortcut(algo, c, h, w):
    """The decoder's ConvTranspose2d -> cat(up, skip) -> 1x1 shortcut as ONE shuffle GEMM with the skip tensor as a
    second, output-resolution source (weights folded in float64 as engine.py does)."""
    g = torch.Generator().manual_seed(21 + c + w)
    N = 2
    cur = torch.randn(N, 2 * c, h, w, generator=g)
    skip = torch.randn(N, c, 2 * h, 2 * w, generator=g)
    wt = torch.randn(2 * c, c, 2, 2, generator=g) / (2 * c) ** 0.5
    bt = torch.randn(c, generator=g)
    wsc = torch.randn(c, 2 * c, 1, 1, generator=g) / (2 * c) ** 0.5
    bsc = torch.randn(c, generator=g)
    ref = F.conv2d(torch.cat([F.conv_transpose2d(cur.double(), wt.double(), bt.double(), stride=2), skip.double()], 1),
                   wsc.double(), bsc.double())
    w2 = wsc.double()[:, :, 0, 0]
    w_cur = torch.einsum('ou,iuyx->ioyx', w2[:, :c], wt.double())
    w_skip = w2[:, c:].t()[:, :, None, None].expand(c, c, 2, 2)
    w_f = torch.cat([w_cur, w_skip], 0).float().contiguous()
    b_f = (bsc.double() + w2[:, :c] @ bt.double()).float()
    got = nchw(run_conv(w_f, b_f, 1, 1, [2 * c, c], [nhwc(cur).to(DEV), nhwc(skip).to(DEV)], N, h, w, shuffle=True, algo=algo))
    assert report(f"fused convT + cat + 1x1 shortcut algo {algo}", got, ref) < 2e-5


def test_channel_padding_small_nf():
    """nf=8 style channel counts are zero-padded to 32 inside the engine."""
    g = torch.Generator().manual_seed(3)
    N, C, H, W = 1, 8, 16, 32
    x = torch.randn(N, C, H, W, generator=g)
    xp = torch.zeros(N, 32, H, W)
    xp[:, :C] = x
    w = torch.randn(16, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(16, generator=g)
    got = nchw(run_conv(w, b, 3, 1, [C], [nhwc(xp).to(DEV)], N, H, W))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    assert report("padded conv", got[:, :16], ref) < 2e-5
    assert float(got[:, 16:].abs().max()) == 0.0


def test_conv_in_and_out_and_maxpool_and_film():
    import ctypes as C
    from yond_public_amd import _lib as L
    from yond_public_amd.engine import DenoiserPlan
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    N, H, W = 2, 24, 40
    x = torch.rand(N, 4, H, W, generator=g)
    ub = x.reshape(N, -1).max(dim=1).values
    w = torch.randn(32, 4, 3, 3, generator=g) / 6
    b = torch.randn(32, generator=g)
    plan = DenoiserPlan.__new__(DenoiserPlan)
    plan.lib, plan.dev = lib, torch.device(DEV)
    wp, bp = plan._pack_conv_in(w, b)
    x4 = nhwc(x).to(DEV)
    ubd = plan.image_max(x4, N)
    torch.cuda.synchronize()
    assert torch.equal(ubd.cpu(), ub)
    dst = torch.empty(N, H, W, 32, device=DEV)
    L.check(lib.yond_conv_in_f32(L.ptr(x4), L.ptr(ubd), N, H, W, 32, L.ptr(wp), L.ptr(bp), 0.01, L.ptr(dst), 0, L.stream()), "conv_in")
    dst4 = torch.empty(N * 32 * H * W, device=DEV)
    L.check(lib.yond_conv_in_f32(L.ptr(x4), L.ptr(ubd), N, H, W, 32, L.ptr(wp), L.ptr(bp), 0.01, L.ptr(dst4), 2, L.stream()), "conv_in")
    assert torch.equal(from_p4(dst4, N, H, W, 32), dst.cpu())          # planes of 4 channels: the same values
    ref = F.leaky_relu(F.conv2d((x / ub.view(-1, 1, 1, 1)).double(), w.double(), b.double(), padding=1), 0.01)
    assert report("conv_in", nchw(dst.cpu()), ref) < 1e-5
    # conv_out
    feat = torch.randn(N, 32, H, W, generator=g)
    w10, b10 = torch.randn(4, 32, 1, 1, generator=g) / 6, torch.randn(4, generator=g)
    out = torch.empty(N, H, W, 4, device=DEV)
    featd, w10d, b10d = nhwc(feat).to(DEV), w10.reshape(4, 32).contiguous().to(DEV), b10.to(DEV)   # keep alive across the launch
    L.check(lib.yond_conv_out_f32(L.ptr(featd), 32, L.ptr(w10d), L.ptr(b10d),
                                  L.ptr(x4), L.ptr(ubd), N, H, W, L.ptr(out), L.stream()), "conv_out")
    refo = (F.conv2d(feat.double(), w10.double(), b10.double()) + (x / ub.view(-1, 1, 1, 1)).double()) * ub.view(-1, 1, 1, 1).double()
    assert report("conv_out", nchw(out.cpu()), refo) < 1e-5
    # maxpool
    mp = torch.empty(N, H // 2, W // 2, 32, device=DEV)
    L.check(lib.yond_maxpool2_f32(L.ptr(dst), N, H, W, 32, L.ptr(mp), L.stream()), "maxpool")
    assert torch.equal(nchw(mp.cpu()), F.max_pool2d(nchw(dst.cpu()), 2))


@pytest.mark.parametrize("c,h,w", [(32, 24, 40), (64, 16, 32)])
def test_decoder_gemm_two_subpositions_per_tile_bit_identical(c, h, w):
    """YondConvDesc.shuffle 2 (the decoder GEMM with two sub-positions per 64-wide tile, K = [cur | skip dx0 | skip dx1 | 0]):
    the same bits as the one-sub-position form (shuffle 1) on split-plane inputs -- products with zero weights add exact zeros."""
    import torch
    from yond_public_amd import engine as E
    from yond_public_amd import _lib as L
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(c + h)
    w_f = (torch.randn((3 * c, c, 2, 2), generator=g) * 0.1)
    b_f = torch.randn(c, generator=g) * 0.1
    plan = E.DenoiserPlan.__new__(E.DenoiserPlan)
    plan.lib, plan.dev = L.load(), dev
    one = E._PackedConv(dev, w_f, b_f, 1, 1, [2 * c, c], shuffle=True)
    two = E._PackedUpSub2(dev, w_f, b_f, c)
    assert two.ok and one.split(2) is not None
    N = 1
    # split-plane inputs: produced by a 3x3 identity-free route is overkill here -- build them on the host from float32 tensors
    def to_sp(x):                                      # x [N][H][W][C] float32 -> split planes [N][C/16][2][2][units][8 halves]
        n_, H, W, C_ = x.shape
        hpart = x.to(torch.float16)
        lpart = ((x - hpart.to(torch.float32)) * 2048.0).to(torch.float16)
        units = E.sp_plane_units(H, W)
        out = torch.zeros((n_, C_ // 16, 2, 2, units, 8), dtype=torch.float16)
        for part, t in enumerate((hpart, lpart)):
            v = t.reshape(n_, H * W, C_ // 16, 2, 8).permute(0, 2, 3, 1, 4)        # [n][c16][half][pixel][8]
            out[:, :, :, part, :H * W, :] = v
        return out.view(torch.float32).reshape(-1).contiguous().to(dev)
    cur = torch.randn((N, h, w, 2 * c), generator=g)
    skip = torch.randn((N, 2 * h, 2 * w, c), generator=g)
    cur_sp, skip_sp = to_sp(cur), to_sp(skip)
    outs = []
    for pc in (one, two):
        dst = torch.full((N, 2 * h, 2 * w, c), float('nan'), device=dev)
        plan._conv(pc, cur_sp, skip_sp, N, h, w, dst, in_fmt=1, algo='split')
        torch.cuda.synchronize()
        outs.append(dst.cpu())
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0], outs[1])
    # ... and both equal the float64 layer to split precision
    ref = torch.einsum('nyxi,iojk->nyjxko', cur.double(), w_f[:2 * c].double()).reshape(N, 2 * h, 2 * w, c)
    ref += torch.einsum('nyjxki,iojk->nyjxko', skip.double().reshape(N, h, 2, w, 2, c), w_f[2 * c:].double()).reshape(N, 2 * h, 2 * w, c)
    ref += b_f.double()
    assert float((outs[1].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("N,H,W,in_nhwc,out4", [(2, 50, 75, False, False), (1, 37, 70, True, True), (1, 24, 64, False, False), (2, 13, 33, True, False),
                                                 (1, 100, 36, False, True), (1, 130, 200, False, False)])
def test_block0_fused_equals_the_two_launches(N, H, W, in_nhwc, out4):
    """csrc/block0_fused.hip: a level-0 residual block (32 -> 32 -> 32 channels) in ONE launch -- conv1 on the tile's halo region, SiLU(FiLM(.))
    split into LDS, conv2 from there, residual, split-plane store or the fused output projection -- against the two split-operand launches
    it replaces (same split arithmetic, another summation order inside the MFMA: agreement to float32 rounding) and against the float64
    block (archs/modules.py:186-196); ragged tiles (H, W no multiples of 12 / 32), batch, both input formats, the zero pads untouched."""
    from yond_public_amd.engine import _PackedConv, DenoiserPlan
    from yond_public_amd import _lib as L
    if not L.has("yond_block0_fused_f32"):
        pytest.skip("the fused level-0 block (a measured no-go) exists only in experiment builds (python -m yond_public_amd.build --experiments)")
    C = 32
    g = torch.Generator().manual_seed(7 * H + W)
    x = torch.randn(N, C, H, W, generator=g)
    w1 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    w2 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    f = [torch.randn(N, C, generator=g).to(DEV) for _ in range(4)]
    xn = nhwc(x)
    xd = xn.to(DEV)
    xin = xd if in_nhwc else to_p4(xn)
    plan = DenoiserPlan.__new__(DenoiserPlan)
    plan.lib, plan.dev = L.load(), torch.device(DEV)
    plan.status, plan.status_slot = torch.zeros(4, dtype=torch.int32, device=DEV), 0
    pc1 = _PackedConv(plan.dev, w1, None, 3, 1, [C])
    pc2 = _PackedConv(plan.dev, w2, None, 3, 1, [C])
    # the two launches: conv1 -> split planes -> conv2 (+ residual)
    t_sp = plan._new_sp('t', N, H, W, C)
    plan._conv(pc1, xd, None, N, H, W, t_sp, escale=f[0], eshift=f[1], ebatch=1, pre_act=1, post_act=1, algo='split', out_fmt=1)
    kw2 = dict(escale=f[2], eshift=f[3], ebatch=1, res=xd, algo='split', in_fmt=1)
    z = F.conv2d(F.silu(x.double()), w1.double(), padding=1) * f[0].cpu().double()[:, :, None, None] + f[1].cpu().double()[:, :, None, None]
    ref = F.conv2d(F.silu(z), w2.double(), padding=1) * f[2].cpu().double()[:, :, None, None] + f[3].cpu().double()[:, :, None, None] + x.double()
    if out4:
        w4 = (torch.randn(4, C, generator=g) / C ** 0.5).to(DEV)
        b4 = torch.randn(4, generator=g).to(DEV)
        x4 = torch.rand(N, H, W, 4, generator=g).to(DEV)
        ub = (torch.rand(N, generator=g) + 0.5).to(DEV)
        o_two = torch.full((N, H, W, 4), float('nan'), device=DEV)
        o_one = torch.full((N, H, W, 4), float('nan'), device=DEV)
        plan._conv(pc2, t_sp, None, N, H, W, None, out4=(w4, b4, x4, ub, o_two), **kw2)
        plan._block0(pc1, pc2, xin, N, H, W, f, 0 if in_nhwc else 2, out4=(w4, b4, x4, ub, o_one))
        torch.cuda.synchronize()
        u = ub.cpu().double()[:, None, None, None]
        ref4 = (torch.einsum('nchw,qc->nhwq', ref, w4.cpu().double()) + b4.cpu().double() + x4.cpu().double() / u) * u
        assert report(f"fused level-0 block + output projection {H}x{W} vs two launches", o_one.cpu(), o_two.cpu()) <= 4e-6 * float(ref4.abs().max())
        assert report(f"fused level-0 block + output projection {H}x{W} vs float64", o_one.cpu(), ref4) <= 6e-5 * max(1.0, float(ref4.abs().max()))
    else:
        o_two = torch.full((N, H, W, C), float('nan'), device=DEV)
        plan._conv(pc2, t_sp, None, N, H, W, o_two, **kw2)
        o_sp = plan._new_sp('o', N, H, W, C)
        plan._block0(pc1, pc2, xin, N, H, W, f, 0 if in_nhwc else 2, dst=o_sp)
        torch.cuda.synchronize()
        val, pads = sp_decode(o_sp, N, C, H, W)
        assert not pads.view(torch.int16).any()
        scale = float(ref.abs().max())
        assert report(f"fused level-0 block {H}x{W} vs two launches", val, nchw(o_two.cpu()).double()) <= 4e-6 * scale
        assert report(f"fused level-0 block {H}x{W} vs float64", val, ref) <= 6e-5 * max(1.0, scale)
    assert int(plan.status[0]) == 0
    # the range guard: an input beyond fp16's range after SiLU is reported
    big = xin.clone()
    big.view(-1)[5] = 1e6
    plan._block0(pc1, pc2, big, N, H, W, f, 0 if in_nhwc else 2, dst=plan._new_sp('o2', N, H, W, C))
    assert int(plan.status[0]) & 1


# ---- the split-plane data flow on H-ONLY planes (descriptor algo 4 with in_fmt / out_fmt: the fp16 path, BASELINE cfg 5) -------------------
def to_hp(x):
    """[N][H][W][C] float32 (CPU) -> h-only planes [N][C/16][channel half][units][8 halves] (float32-typed buffer on the device): the halves
    the plain-tensor h-only kernels round when they stage."""
    from yond_public_amd.engine import sp_plane_units
    N, H, W, C = x.shape
    ps = sp_plane_units(H, W)
    out = torch.zeros(N, C // 16, 2, ps, 8, dtype=torch.float16)
    out[:, :, :, :H * W, :] = x.half().reshape(N, H * W, C // 16, 2, 8).permute(0, 2, 3, 1, 4)
    return out.reshape(-1).view(torch.float32).to(DEV)


def hp_decode(sp, N, C, H, W):
    """h-only planes -> ([N][H][W][C] float16 (CPU), the pad units)."""
    from yond_public_amd.engine import sp_plane_units
    u = sp.cpu().view(torch.float16).reshape(N, C // 16, 2, sp_plane_units(H, W), 8)
    return u[:, :, :, :H * W, :].permute(0, 3, 1, 2, 4).reshape(N, H, W, C), u[:, :, :, H * W:, :]


@pytest.mark.parametrize("C,N,H,W", [(64, 2, 40, 70), (128, 1, 380, 100), (32, 2, 50, 75), (256, 1, 30, 61), (64, 5, 16, 16), (32, 1, 37, 70),
                                     # enough tiles for the four-rows-per-wave forms (16 x 32 pixels x 64 channels, 32 x 32 x 32)
                                     (64, 1, 260, 520), (32, 1, 520, 530)])
def test_half_plane_flow_block(C, N, H, W):
    """The residual block in the formats of the fp16 path's data flow (round 6): x in float32 planes of 4 channels (conv1's staged input,
    conv2's residual), tmp and out in H-ONLY planes (2 bytes per element).  The plain-tensor h-only kernels round every operand to half when
    they stage it; here the producer stores the rounded value instead: tmp and out must be the half-rounding of what the [N][H][W][C] h-only
    path computes, bit for bit, the zero pads untouched; every tile shape (8 / 12 / 16-row tiles, partial tiles, batch)."""
    from yond_public_amd.engine import _PackedConv
    g = torch.Generator().manual_seed(C + H)
    x = nhwc(torch.randn(N, C, H, W, generator=g))
    w1 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    w2 = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    es, et = torch.randn(N, C, generator=g).to(DEV), torch.randn(N, C, generator=g).to(DEV)
    es2, et2 = torch.randn(N, C, generator=g).to(DEV), torch.randn(N, C, generator=g).to(DEV)
    plan = bare_plan()
    plan.status, plan.status_slot = torch.zeros(4, dtype=torch.int32, device=DEV), 0
    pc1, pc2 = _PackedConv(plan.dev, w1, None, 3, 1, [C]), _PackedConv(plan.dev, w2, None, 3, 1, [C])
    xd, xp = x.to(DEV), to_p4(x)
    kw1 = dict(escale=es, eshift=et, ebatch=1, pre_act=1, post_act=1, algo='half')
    kw2 = dict(escale=es2, eshift=et2, ebatch=1, algo='half')
    t_ref = torch.empty(N, H, W, C, device=DEV)
    o_ref = torch.empty(N, H, W, C, device=DEV)
    plan._conv(pc1, xd, None, N, H, W, t_ref, **kw1)
    plan._conv(pc2, t_ref, None, N, H, W, o_ref, res=xd, **kw2)
    t_hp, o_hp = plan._new_sp('t', N, H, W, C, 1), plan._new_sp('o', N, H, W, C, 1)
    plan._conv(pc1, xp, None, N, H, W, t_hp, in_fmt=2, out_fmt=1, **kw1)
    plan._conv(pc2, t_hp, None, N, H, W, o_hp, res=xp, in_fmt=1, out_fmt=1, res_fmt=2, **kw2)
    torch.cuda.synchronize()
    for name, sp, ref in (("tmp", t_hp, t_ref), ("out", o_hp, o_ref)):
        h, pads = hp_decode(sp, N, C, H, W)
        assert not pads.view(torch.int16).any(), name
        assert torch.equal(h.view(torch.int16), ref.cpu().half().view(torch.int16)), (name, float((h.float() - ref.cpu()).abs().max()))
    assert int(plan.status[0]) == 0
    z = F.conv2d(F.silu(nchw(x).double()), w1.double(), padding=1) * es.cpu().double()[:, :, None, None] + et.cpu().double()[:, :, None, None]
    ref = F.conv2d(F.silu(z), w2.double(), padding=1) * es2.cpu().double()[:, :, None, None] + et2.cpu().double()[:, :, None, None] + nchw(x).double()
    got = nchw(hp_decode(o_hp, N, C, H, W)[0].float())
    assert report(f"residual block through h-only planes C{C} {H}x{W}", got, ref) <= 0.02 * max(1.0, float(ref.abs().max()))
    # the last block's form: conv2 with the fused output projection reads tmp in h-only planes, its residual as [N][H][W][C]
    if C == 32:
        w4 = (torch.randn(4, C, generator=g) / C ** 0.5).to(DEV)
        b4 = torch.randn(4, generator=g).to(DEV)
        xin = torch.rand(N, H, W, 4, generator=g).to(DEV)
        ub = (torch.rand(N, generator=g) + 0.5).to(DEV)
        o4 = torch.full((N, H, W, 4), float('nan'), device=DEV)
        plan._conv(pc2, t_hp, None, N, H, W, None, res=xd, in_fmt=1, out4=(w4, b4, xin, ub, o4), **kw2)
        torch.cuda.synchronize()
        u = ub.cpu().double()[:, None, None, None]
        want = (torch.einsum('nhwc,qc->nhwq', o_ref.cpu().double(), w4.cpu().double()) + b4.cpu().double() + xin.cpu().double() / u) * u
        assert report("h-only conv2 + fused output projection", o4.cpu(), want) <= 2e-5 * max(1.0, float(want.abs().max()))
    # the range guard: a stored value beyond fp16's range is reported
    big = x.clone()
    big.view(-1)[3] = 3e6
    plan._conv(pc1, to_p4(big), None, N, H, W, plan._new_sp('t2', N, H, W, C, 1), in_fmt=2, out_fmt=1, **kw1)
    assert int(plan.status[0]) & 1


@pytest.mark.parametrize("C,N,H,W", [(32, 2, 37, 70), (64, 1, 64, 130), (128, 1, 23, 45), (256, 1, 12, 33),
                                     # enough tiles for the 8-row forms (two output rows per wave): 64-channel tiles, and 128-channel tiles where there is no second output
                                     (32, 2, 250, 510), (64, 2, 256, 500), (128, 1, 130, 260)])
def test_conv3x3_s2_half_planes(C, N, H, W):
    """Stride-2 layer of the fp16 path's flow: h-only planes in (through the register sets), float32 planes of 4 channels out -- the
    plain-tensor h-only kernel's result bit for bit; the second output (YondConvDesc.dst2) holds half(SiLU(value)) in h-only planes."""
    from yond_public_amd.engine import _PackedConv
    g = torch.Generator().manual_seed(C + W)
    x = nhwc(torch.randn(N, C, H, W, generator=g))
    w = torch.randn(2 * C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(2 * C, generator=g)
    plan = bare_plan()
    pc = _PackedConv(plan.dev, w, b, 3, 2, [C])
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    ref = torch.empty(N, Ho, Wo, 2 * C, device=DEV)
    plan._conv(pc, x.to(DEV), None, N, H, W, ref, algo='half')
    gp4 = torch.empty(N * 2 * C * Ho * Wo, device=DEV)
    plan._conv(pc, to_hp(x), None, N, H, W, gp4, algo='half', in_fmt=1, out_fmt=2)
    gp4b = torch.empty(N * 2 * C * Ho * Wo, device=DEV)
    second = plan._new_sp('second', N, Ho, Wo, 2 * C, 1)
    plan._conv(pc, to_hp(x), None, N, H, W, gp4b, algo='half', in_fmt=1, out_fmt=2, dst2=second)
    torch.cuda.synchronize()
    assert torch.equal(from_p4(gp4, N, Ho, Wo, 2 * C), ref.cpu())
    assert torch.equal(gp4b.cpu(), gp4.cpu())
    val, pads = hp_decode(second, N, 2 * C, Ho, Wo)
    assert not pads.view(torch.int16).any()
    want = F.silu(ref.cpu().double())
    assert float((val.double() - want).abs().max()) <= 2.0 ** -10 * max(1.0, float(want.abs().max()))
    z = F.conv2d(nchw(x.half().float()).double(), w.half().double(), b.double(), stride=2, padding=1)
    assert report(f"stride-2 from h-only planes C{C}", nchw(ref.cpu()), z) <= 2e-5 * max(1.0, float(z.abs().max()))


@pytest.mark.parametrize("c,h,w,N", [(64, 24, 40, 2), (32, 19, 33, 2), (128, 9, 35, 2), (256, 7, 20, 1),
                                     # enough tiles for the 16-row form at 64 columns
                                     (64, 200, 330, 1)])
def test_decoder_gemm_half_planes(c, h, w, N):
    """The decoder GEMM of the fp16 path's flow (ConvTranspose2d 2x2 + cat + 1x1 shortcut folded; archs/Unet.py:447-461): BOTH sources in
    h-only planes, output as float32 planes of 4 channels or [N][H][W][C] (the last block's input) -- against the float64 layer on the
    half-rounded operands (fp32 accumulate: 2e-5); the two output formats bit-equal; the second output = half(SiLU(value)); at 32 channels
    the two-sub-positions-per-tile form (YondConvDesc.shuffle 2) bit-equal to the plain form."""
    from yond_public_amd.engine import _PackedConv, _PackedUpSub2
    g = torch.Generator().manual_seed(c + h)
    cur = nhwc(torch.randn(N, 2 * c, h, w, generator=g))
    skip = nhwc(torch.randn(N, c, 2 * h, 2 * w, generator=g))
    wf = torch.randn(3 * c, c, 2, 2, generator=g) / (3 * c) ** 0.5        # ConvTranspose2d layout over [cur | skip]
    bf = torch.randn(c, generator=g)
    plan = bare_plan()
    pc = _PackedConv(plan.dev, wf, bf, 1, 1, [2 * c, c], shuffle=True)
    got = torch.empty(N, 2 * h, 2 * w, c, device=DEV)
    plan._conv(pc, to_hp(cur), to_hp(skip), N, h, w, got, algo='half', in_fmt=1)
    gp4 = torch.empty(N * c * 4 * h * w, device=DEV)
    plan._conv(pc, to_hp(cur), to_hp(skip), N, h, w, gp4, algo='half', in_fmt=1, out_fmt=2)
    torch.cuda.synchronize()
    wq = wf.half().double()
    ref = torch.einsum('nyxi,iojk->nyjxko', cur.half().double(), wq[:2 * c]).reshape(N, 2 * h, 2 * w, c)
    ref += torch.einsum('nyjxki,iojk->nyjxko', skip.half().double().reshape(N, h, 2, w, 2, c), wq[2 * c:]).reshape(N, 2 * h, 2 * w, c)
    ref += bf.double()
    assert report(f"decoder GEMM from h-only planes c{c}", got.cpu(), ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(from_p4(gp4, N, 2 * h, 2 * w, c), got.cpu())
    if c >= 64:
        gp4b = torch.empty(N * c * 4 * h * w, device=DEV)
        second = plan._new_sp('second', N, 2 * h, 2 * w, c, 1)
        plan._conv(pc, to_hp(cur), to_hp(skip), N, h, w, gp4b, algo='half', in_fmt=1, out_fmt=2, dst2=second)
        torch.cuda.synchronize()
        assert torch.equal(gp4b.cpu(), gp4.cpu())
        val, pads = hp_decode(second, N, c, 2 * h, 2 * w)
        assert not pads.view(torch.int16).any()
        want = F.silu(got.cpu().double())
        assert float((val.double() - want).abs().max()) <= 2.0 ** -10 * max(1.0, float(want.abs().max()))
    if c == 32:
        up2 = _PackedUpSub2(plan.dev, wf, bf, c)
        assert up2.ok and up2.split(1) is not None
        g2 = torch.empty(N * c * 4 * h * w, device=DEV)
        plan._conv(up2, to_hp(cur), to_hp(skip), N, h, w, g2, algo='half', in_fmt=1, out_fmt=2)
        torch.cuda.synchronize()
        assert report("two sub-positions per tile, h-only", from_p4(g2, N, 2 * h, 2 * w, c), got.cpu()) <= 2e-6 * max(1.0, float(ref.abs().max()))
