"""GPU: IterDenoise (row Q) against the reference's captured outputs, metrics kernel, full-frame smoke."""
import numpy as np
import pytest
import torch

from hip_common import ARCHS, make_net, report, sha

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize("ci", range(8))
def test_iter_denoise_matches_reference(golden, ci):
    """Row Q against the reference's own IterDenoise (tests/golden/iter.npz, oracle/gen_golden.py ITER_CASES): cases 0-1
    end at the beta1 < 0 guard (YOND_SIDD.py:445-447, one output), 2-3 take the beta2 < 0 -> beta1**2 branch (:438-440)
    and continue, 4-7 run the plain second round (second get_bias :450-454, second K1 -> net -> K4 pass :456-467);
    block-wise (batch 32) and full_dn, GuidedResUnet nf 8 / 32 and UNetSeeInDark."""
    from test_oracle_golden import iter_case, iter_crop
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    g = golden("iter")
    lr, clean, full, arch, sd, pipe = iter_case(g, ci)
    net = getattr(A, arch['name'])(dict(arch))
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    res = P.IterDenoise(lr, net, arch, pipe, lr_full=full, device=DEV)
    regs = g[f"regs_{ci}"]
    assert len(res['regs']) == len(regs) and len(res['raw_dns']) == int(g[f"nout_{ci}"])
    for r, gr in zip(res['regs'], regs):
        print(f"[parity] case {ci} regs {r[0]:.6e},{r[1]:.6e} vs {gr[0]:.6e},{gr[1]:.6e}")
        np.testing.assert_allclose(r[0], gr[0], rtol=2e-5)
        np.testing.assert_allclose(r[1], gr[1], rtol=0, atol=2e-5 * abs(gr[0]) + 1e-9)
    for it, dn in enumerate(res['raw_dns']):
        dn = dn.cpu().numpy()
        for got, tag in zip(iter_crop(dn), ("blk", "seam", "sub")):
            assert report(f"IterDenoise case {ci} iter {it} {tag}", got, g[f"dn_{ci}_{it}_{tag}"]) <= 1e-4
        np.testing.assert_allclose(np.asarray(dn, np.float64).sum(), g[f"dn_{ci}_{it}_chk"][0], rtol=1e-5)


def test_iter_denoise_bare_full_frame_matches_reference(golden):
    """A bare full-frame 'iter' run (full_dn, not the SIDD stack) against the reference's own IterDenoise on a 320 x 2048
    frame (tests/golden/iter_full.npz): the packed width divides by 32, so the reference's hard-coded SIDD_256 re-tiling of
    the collaborative estimate (YOND_SIDD.py:431, :91-93) runs -- and is what pipeline.IterDenoise does by default there."""
    from test_oracle_golden import iter_full_case, iter_full_crop
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    g = golden("iter_full")
    noisy, clean, arch, sd, pipe = iter_full_case()
    assert np.array_equal(sha(noisy), g["sha"])
    net = A.GuidedResUnet(dict(arch))
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    res = P.IterDenoise(noisy, net, arch, pipe, device=DEV)
    assert len(res['raw_dns']) == int(g["nout"]) == 2
    for r, gr in zip(res['regs'], g["regs"]):
        print(f"[parity] bare full frame regs {r[0]:.6e},{r[1]:.6e} vs {gr[0]:.6e},{gr[1]:.6e}")
        np.testing.assert_allclose(r[0], gr[0], rtol=2e-5)
        np.testing.assert_allclose(r[1], gr[1], rtol=0, atol=2e-5 * abs(gr[0]) + 1e-9)
    for it, dn in enumerate(res['raw_dns']):
        for got, tag in zip(iter_full_crop(dn.cpu().numpy()), "abc"):
            assert report(f"bare full-frame IterDenoise iter {it} {tag}", got, g[f"dn_{it}_{tag}"]) <= 1e-4


def test_device_chain_equals_host_chain():
    """The per-frame parameter chain on the device (csrc/frame_chain.hip: moments -> beta -> K, sigma, lower, upper, t, knot grid
    -> bias LUT -> K1 -> net -> K4, no host round trip) against the host-side chain it replaces, on the same frames: the
    estimates agree to float64 rounding, the knot grids and LUT ordinates bit for bit, the outputs to float32 rounding."""
    import yond_oracle as O
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    from yond_public_amd import synthetic as S
    arch = ARCHS["gru8"]
    net = A.GuidedResUnet(dict(arch))
    net.load_state_dict(S.denoising_state_dict(net, 3))
    net = net.to(DEV).eval()
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': True}
    for (H, W, K, s, clipf, expo) in [(512, 768, 4.0, 6.0, True, 1.0), (384, 640, 2.0, 20.0, True, 1.0), (320, 512, 1.0, 12.0, False, 0.2)]:
        noisy, _ = O.synth_noisy(H, W, K, s, 11, clip=clipf)
        x = torch.from_numpy((noisy * expo).astype(np.float32)).to(DEV)
        assert P.chain_applies(x, net, arch, pipe)
        res_d = P.IterDenoise(x, net, arch, pipe)
        lut_d = [P._chain_buffers(x.device, sl) for sl in (0, 1)]
        knots_d = [b.lut_x.cpu().numpy()[:int(b.prm_host[P.PRM['lut_n']])].copy() for b in lut_d]
        ys_d = [b.lut_y.cpu().numpy()[:int(b.prm_host[P.PRM['lut_n']])].copy() for b in lut_d]
        P.DEVICE_CHAIN = False
        try:
            res_h = P.IterDenoise(x, net, arch, pipe)
        finally:
            P.DEVICE_CHAIN = True
        assert len(res_d['raw_dns']) == len(res_h['raw_dns'])
        for rd, rh in zip(res_d['regs'], res_h['regs']):
            # (the moment sums are float64 atomics: their order, and with it the last bits of the fit, differs from run to run)
            np.testing.assert_allclose(rd[0], rh[0], rtol=1e-11)
            np.testing.assert_allclose(rd[1], rh[1], rtol=1e-10, atol=1e-20)
        for pd_, ph in zip(res_d['params'], res_h['params']):
            np.testing.assert_allclose(pd_, ph, rtol=1e-10)
        for it, (a, b) in enumerate(zip(res_d['raw_dns'], res_h['raw_dns'])):
            assert report(f"device chain vs host chain {H}x{W} round {it}", a.cpu().numpy(), b.cpu().numpy()) <= 2e-7
        # knot grid and ordinates of each round that ran: the reference's own np.linspace calls / the host-launched LUT kernel
        lr_max = np.float32(res_d['nle_info']['frame_max'])
        for it in range(len(res_d['raw_dns'])):
            K_, s_ = res_d['params'][it]
            f = P.get_bias(lr_max * np.float32(959.0), s_, K_, device=x.device)
            assert np.array_equal(knots_d[it], np.asarray(f.lams, np.float64))
            assert np.array_equal(ys_d[it], f.y.cpu().numpy())


@pytest.mark.parametrize("with_full", [False, True])
def test_sidd_layout_device_chain_equals_host_chain(with_full):
    """The SIDD layout ([32][H][W] blocks, block-wise denoising, round 1's estimate on the blocks' concatenation or on a separate full
    frame, YOND_SIDD.py:312-470) on the device chain -- batched K1 / K4 reading the parameter block, no host round trip between the
    rounds -- against the host-side chain: estimates to the order of the float64 atomics, outputs to 2e-7, and the path really taken."""
    import yond_oracle as O
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    from yond_public_amd import synthetic as S
    arch = ARCHS["gru8"]
    net = A.GuidedResUnet(dict(arch))
    net.load_state_dict(S.denoising_state_dict(net, 3))
    net = net.to(DEV).eval()
    noisy, _ = O.synth_noisy(64, 32 * 96, 4.0, 6.0, 21)
    blocks = torch.from_numpy(np.stack(np.split(noisy, 32, axis=-1))).to(DEV)
    full = torch.from_numpy(O.synth_noisy(320, 512, 4.0, 6.0, 22)[0]).to(DEV) if with_full else None
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': False}
    p = P.default_params()
    assert P.chain_applies_sidd(blocks, full, net, arch, pipe, p)
    calls = []
    orig = P._iter_denoise_chain_sidd
    P._iter_denoise_chain_sidd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        res_d = P.IterDenoise(blocks, net, arch, pipe, lr_full=full)
    finally:
        P._iter_denoise_chain_sidd = orig
    assert calls and 'nle_info' in res_d                        # (the device chain ran and was not handed back to the host path)
    P.CHAIN_SIDD = False
    try:
        res_h = P.IterDenoise(blocks, net, arch, pipe, lr_full=full)
    finally:
        P.CHAIN_SIDD = True
    assert len(res_d['raw_dns']) == len(res_h['raw_dns']) >= 1          # (round 2 may end at the reference's beta1 < 0 guard: on both paths alike)
    print(f"[parity] SIDD layout: {len(res_d['raw_dns'])} round(s), regs {res_d['regs']}")
    for rd, rh in zip(res_d['regs'], res_h['regs']):
        np.testing.assert_allclose(rd[0], rh[0], rtol=1e-10)
        np.testing.assert_allclose(rd[1], rh[1], rtol=1e-9, atol=1e-20)
    for it, (a, b) in enumerate(zip(res_d['raw_dns'], res_h['raw_dns'])):
        assert a.shape == b.shape == (64, 32 * 96)
        assert report(f"SIDD layout, device chain vs host chain, round {it}", a.cpu().numpy(), b.cpu().numpy()) <= 2e-7


@pytest.mark.parametrize("ub", [3.0, 17.0, 49.0, 50.0, 51.0, 137.0, 499.0, 500.0, 501.0, 509.0, 510.0, 961.0, 1024.0, 1475.0])
def test_device_knot_grid_bit_identical(ub):
    """yond_frame_params_f64's knot grid against the reference's np.linspace calls (utils/isp_algos.py:101-108, pipeline._bias_knots)
    for float32 maxima on both sides of every branch: bit-identical float64 knots, NumPy 2's float32 linspace included."""
    import ctypes as C
    from yond_public_amd import _lib as L
    from yond_public_amd import pipeline as P
    lib = L.load()
    mx = np.float32(ub - 1.0) - np.float32(0.3)                 # ceil(mx) + 1 == ub
    assert np.ceil(mx) + 1 == np.float32(ub)
    ws = torch.zeros(int(lib.yond_nle_ws_bytes(1024)), dtype=torch.uint8, device=DEV)
    off_mom = P._nle_layout()[2]
    mom = np.array([100, 30, 0.3, 10, 0.1, 100, 30, 0.3, 10, 0.1], np.float64)      # any well-posed moment sums
    ws[off_mom:off_mom + 80] = torch.from_numpy(mom.view(np.uint8)).to(DEV)
    mxd = torch.tensor([float(mx) / 959.0], dtype=torch.float32, device=DEV)
    # (the kernel multiplies the float32 maximum by float32(scale): feed it scale = 1 and the maximum itself)
    mxd = torch.tensor([float(mx)], dtype=torch.float32, device=DEV)
    prm = torch.zeros(16, dtype=torch.float64, device=DEV)
    lut_x = torch.zeros(P.LUT_CAP, dtype=torch.float64, device=DEV)
    L.check(lib.yond_frame_params_f64(L.ptr(ws), L.ptr(mxd), 0, 1.0, 1.0, 1.03, P.LUT_CAP, L.ptr(prm), None, L.ptr(lut_x), L.stream()), "frame_params")
    torch.cuda.synchronize()
    n = int(prm[P.PRM['lut_n']].item())
    want = np.asarray(P._bias_knots(np.ceil(mx) + 1), np.float64)
    assert n == len(want)
    assert np.array_equal(lut_x.cpu().numpy()[:n], want)


@pytest.mark.parametrize("ub,mom,mode", [(961.0, [100, 30, 0.3, 10, 0.1] * 2, 0), (137.0, [5e5, 1.2e5, 900.0, 4.1e4, 310.0, 4e5, 1.0e5, 800.0, 3.5e4, 260.0], 1),
                                         (17.0, [100, 30, 0.3, 10, 0.1] * 2, 0), (2500.0, [100, 30, 0.3, 10, 0.1] * 2, 0),
                                         (961.0, [100, 30, -0.3, 10, -0.1] * 2, 0), (961.0, [0.0] * 10, 0)])
def test_frame_chain_one_launch_equals_three(ub, mom, mode):
    """yond_frame_chain_f64 (parameters + knots + bias LUT + prepared table in one launch) against the three launches it replaces:
    the same parameter block, knots, ordinates and table, bit for bit -- also for the flagged frames (more knots than the capacity,
    K <= 0, no flat area) and when called twice on the same workspace (the arrival counter goes back to zero)."""
    from yond_public_amd import _lib as L
    from yond_public_amd import pipeline as P
    lib = L.load()
    mx = np.float32(ub - 1.0) - np.float32(0.3)
    ws = torch.zeros(int(lib.yond_nle_ws_bytes(1024)), dtype=torch.uint8, device=DEV)
    off_mom = P._nle_layout()[2]
    ws[off_mom:off_mom + 80] = torch.from_numpy(np.array(mom, np.float64).view(np.uint8)).to(DEV)
    mxd = torch.tensor([float(mx) / 959.0], dtype=torch.float32, device=DEV)

    def bufs():
        return (torch.full((16,), -7.0, dtype=torch.float64, device=DEV), torch.zeros(1, dtype=torch.float32, device=DEV),
                torch.zeros(P.LUT_CAP, dtype=torch.float64, device=DEV), torch.zeros(P.LUT_CAP, dtype=torch.float32, device=DEV),
                torch.zeros(int(lib.yond_lut_ws_bytes(P.LUT_CAP)), dtype=torch.uint8, device=DEV))
    prm, t, lx, ly, lw = bufs()
    L.check(lib.yond_frame_params_f64(L.ptr(ws), L.ptr(mxd), mode, 959.0, 959.0, 1.03, P.LUT_CAP, L.ptr(prm), L.ptr(t), L.ptr(lx), L.stream()), "params")
    L.check(lib.yond_bias_lut_dev_f64(L.ptr(lx), P.LUT_CAP, L.ptr(prm), L.ptr(ly), L.stream()), "lut")
    L.check(lib.yond_lut_table_f64(L.ptr(lx), L.ptr(ly), -1, L.ptr(prm), L.ptr(lw), L.stream()), "table")
    for rep in range(2):
        prm2, t2, lx2, ly2, lw2 = bufs()
        L.check(lib.yond_frame_chain_f64(L.ptr(ws), L.ptr(mxd), mode, 959.0, 959.0, 1.03, P.LUT_CAP, L.ptr(prm2), L.ptr(t2), L.ptr(lx2),
                                         L.ptr(ly2), L.ptr(lw2), L.stream()), "chain")
        torch.cuda.synchronize()
        a, b = prm.cpu().numpy(), prm2.cpu().numpy()
        print(f"[parity] frame chain ub {ub}: flags {int(a[P.PRM['flags']])}, {int(a[P.PRM['lut_n']])} knots, K {a[P.PRM['gain']]:.4f}")
        assert np.array_equal(a[:14].view(np.uint64), b[:14].view(np.uint64))
        assert torch.equal(t.view(torch.int32), t2.view(torch.int32)) and torch.equal(lx, lx2) and torch.equal(ly, ly2)     # (t may be NaN)
        n = int(a[P.PRM['lut_n']]) if not (int(a[P.PRM['flags']]) & 10) else 0
        hd, hd2 = lw[:144].cpu().numpy(), lw2[:144].cpu().numpy()
        assert np.array_equal(hd[:16], hd2[:16])                 # n, nseg, nbreak
        if n >= 2:
            ns = int(hd[4:8].view(np.int32)[0])                     # (entries beyond the runs in use are whatever the LDS held)
            for o, cnt in ((16, ns), (48, ns), (80, ns + 1)):
                assert np.array_equal(hd[o:o + 4 * cnt], hd2[o:o + 4 * cnt])
            ab0, x0 = 144, 144 + 4096 * 16
            assert torch.equal(lw[ab0:ab0 + 16 * n], lw2[ab0:ab0 + 16 * n]) and torch.equal(lw[x0:x0 + 8 * n], lw2[x0:x0 + 8 * n])


@pytest.mark.parametrize("H,W,K,sg,expo", [(512, 768, 4.0, 6.0, 1.0), (130, 258, 1.0, 12.0, 0.2), (64, 2100, 9.0, 2.0, 0.04)])
def test_chain_k1_equals_general_k1(H, W, K, sg, expo):
    """yond_pack_vst_norm_chain_f32 (affine tail folded into the table's coefficients, runs in registers, branch-free root) against
    yond_pack_vst_norm_dev_f32 on the same parameter block and table: the float64 value moves by ~1e-16, so after the one rounding to
    float32 at most a handful of elements differ, by one ulp; one, two and three runs of knots (frame maxima below 50, below 500, above)."""
    import yond_oracle as O
    from yond_public_amd import _lib as L
    from yond_public_amd import pipeline as P
    lib = L.load()
    noisy, _ = O.synth_noisy(H, W, K, sg, 7)
    x = torch.from_numpy((noisy * expo).astype(np.float32)).to(DEV)
    x[0, :8] = -0.01                                             # (negative input: the VST argument clamps at zero)
    pipe = {'k': 29}
    p = {'wp': 1023.0, 'bl': 64.0, 'scale': 959.0}
    buf = P._chain_buffers(x.device, 0)
    P._chain_estimate(x, None, 'self', pipe, p, buf)
    h, w = H // 2, W // 2
    p2d = P.get_p2d((1, 4, h, w), base=32)
    Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
    outs, maxes = [], []
    for fn in (lib.yond_pack_vst_norm_dev_f32, lib.yond_pack_vst_norm_chain_f32):
        o = torch.full((Hp, Wp, 4), -1.0, dtype=torch.float32, device=DEV)
        mx = torch.zeros(1, dtype=torch.float32, device=DEV)
        L.check(fn(L.ptr(x), H, W, L.ptr(o), p2d[0], p2d[1], p2d[2], p2d[3], 959.0, L.ptr(buf.prm), L.ptr(buf.lut_ws), P.LUT_CAP, L.ptr(mx),
                   L.stream()), "K1")
        outs.append(o)
        maxes.append(mx)
    torch.cuda.synchronize()
    prm = buf.prm.cpu().numpy()
    assert int(prm[P.PRM['flags']]) == 0
    a, b = outs[0].cpu().numpy(), outs[1].cpu().numpy()
    diff = np.abs(a - b)
    ndiff = int((diff > 0).sum())
    print(f"[parity] chain K1 vs general K1 {H}x{W}: {int(prm[P.PRM['lut_n']])} knots, {ndiff} of {a.size} elements differ, max {diff.max():.2e}")
    assert a.min() >= 0.0 and diff.max() <= 6e-8 and ndiff <= max(2, a.size // 1000000)
    assert abs(float(maxes[0]) - float(maxes[1])) <= 6e-8


def test_block_metrics_vs_oracle():
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    noisy, clean = O.synth_noisy(256, 1024, 4.0, 6.0, 41)
    dn = np.clip(clean + 0.01 * (noisy - clean), 0, 1).astype(np.float32)
    psnr, ssim = P.block_metrics(torch.from_numpy(dn).to(DEV), torch.from_numpy(clean).to(DEV))
    dns, hrs = np.split(dn, 4, axis=-1), np.split(clean, 4, axis=-1)
    rp = np.array([O.psnr(a, b) for a, b in zip(dns, hrs)])
    rs = np.array([O.ssim(a * 255, b * 255) for a, b in zip(dns, hrs)])
    print("[parity] psnr", psnr, rp, "ssim", ssim, rs)
    np.testing.assert_allclose(psnr, rp, rtol=0, atol=1e-6)
    np.testing.assert_allclose(ssim, rs, rtol=0, atol=1e-9)


def test_full_frame_cfg2_properties():
    """BASELINE cfg 2 (3000 x 4000, SNR-Net nf=32, full pipeline, once): properties that do not need the
    oracle at this size -- output in [0,1], finite, denoising reduces the error against the clean frame."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    arch = ARCHS["gru32"]
    net, _ = make_net(arch, 0)
    noisy, clean = O.synth_noisy(3000, 4000, 4.0, 6.0, 0)
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'full_dn': True}
    res = P.IterDenoise(torch.from_numpy(noisy).to(DEV), net, arch, pipe)
    dn = res['raw_dns'][0]
    torch.cuda.synchronize()
    assert dn.shape == (3000, 4000) and bool(torch.isfinite(dn).all())
    assert float(dn.min()) >= 0.0 and float(dn.max()) <= 1.0
    print("[cfg2] params", res['params'])


@pytest.mark.parametrize("K,sigma,expo,idx", [(1.0, 5.0, 0.03, 51), (2.0, 25.0, 0.03, 52), (4.0, 50.0, 0.2, 53)])
def test_cfg5_unclipped_low_light_sweep(K, sigma, expo, idx):
    """BASELINE cfg 5: no black-level clip (negative DN reach the VST; the bias LUT is looked up at max(x, 0),
    YOND_SIDD.py:252), sigma sweep 5..50 DN.  fp32 path against the oracle on the same unclipped frame, then the
    fp16 MFMA conv path against the fp32 output (PSNR >= 55 dB)."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    arch = ARCHS["gru32"]
    net, sd = make_net(arch, 3)
    # low light: the synthetic scene at 3 % / 20 % exposure, Poisson-Gaussian noise, NOT clipped at the black level
    rng = np.random.default_rng(idx)
    clean = (O.synth_clean(192, 256) * expo).astype(np.float32)
    noisy = ((rng.poisson(clean * 959.0 / K) * K + rng.normal(0.0, sigma, clean.shape)) / 959.0).astype(np.float32)
    assert noisy.min() < 0.0                                   # the case really has negative pixels
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'full_dn': True}
    torch.set_num_threads(8)
    ref = O.IterDenoise(noisy, arch, sd, pipe)
    res = P.IterDenoise(noisy, net, arch, pipe, device=DEV)
    np.testing.assert_allclose(res['regs'][0][0], ref['regs'][0][0], rtol=2e-5)
    np.testing.assert_allclose(res['regs'][0][1], ref['regs'][0][1], rtol=0, atol=2e-5 * abs(ref['regs'][0][0]) + 1e-9)
    dn32 = res['raw_dns'][0].cpu().numpy()
    assert report(f"cfg5 K={K} sigma={sigma} fp32 vs oracle", dn32, ref['raw_dns'][0]) <= 1e-4
    net.precision = 'fp16'
    dn16 = P.IterDenoise(noisy, net, arch, pipe, device=DEV)['raw_dns'][0].cpu().numpy()
    mse = float(np.mean((dn16.astype(np.float64) - dn32.astype(np.float64)) ** 2))
    psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
    print(f"[parity] cfg5 K={K} sigma={sigma}: fp16-MFMA path vs fp32 path PSNR {psnr:.1f} dB")
    assert psnr >= 55.0


def test_denoise_stream_matches_iterdenoise():
    """The two-stream driver (NLE of frame k+1 overlapped with the network of frame k) must return, frame by frame,
    what IterDenoise returns (same kernels and arguments; the NLE's float64 moment sums are accumulated with atomics,
    so the estimates agree to rounding of the summation order, not bit for bit -- also between two IterDenoise runs)."""
    import yond_public_amd.pipeline as P
    import yond_public_amd.archs as A
    import yond_public_amd.synthetic as S
    dev = torch.device('cuda:0')
    arch = dict(name='GuidedResUnet', in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True, guided=True)
    net = A.GuidedResUnet(dict(arch))
    net.load_state_dict(S.procedural_state_dict(net, 0))
    net = net.to(dev).eval()
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'max_iter': 1, 'full_dn': True}
    # round 6: the network passes of consecutive frames alternate between two streams (pipeline.STREAM_LANES), each lane with its own split-plane
    # tensors -- frames of two sizes, more of them than the driver's ring of parameter blocks (4) holds, and a one-frame sequence that must drain
    shapes = [(256, 320), (256, 320), (320, 448), (256, 320), (320, 448), (256, 320), (256, 320)]
    frames = [torch.from_numpy(S.synth_noisy(h, w, 3.0 + i, 5.0 + 2 * i, 10 + i)[0]).to(dev) for i, (h, w) in enumerate(shapes)]
    seq = [P.IterDenoise(f, net, arch, pipe) for f in frames]
    assert P.STREAM_LANES == 2
    got = list(P.denoise_stream(iter(frames), net, arch, pipe))
    one = list(P.denoise_stream(iter(frames[:1]), net, arch, pipe))
    assert len(one) == 1 and float((one[0]['raw_dns'][0] - seq[0]['raw_dns'][0]).abs().max()) <= 5e-6
    P.STREAM_LANES = 1                                            # (every pass on the main stream, as before)
    try:
        got1 = list(P.denoise_stream(iter(frames), net, arch, pipe))
    finally:
        P.STREAM_LANES = 2
    for a_, b_ in zip(got1, seq):
        assert float((a_['raw_dns'][0] - b_['raw_dns'][0]).abs().max()) <= 5e-6
    got_host = list(P.denoise_stream((f.cpu().numpy() for f in frames), net, arch, pipe, device=dev))    # host arrays are uploaded in order
    torch.cuda.synchronize()
    assert len(got) == len(seq) == len(got_host)
    for a_, b_ in zip(got_host, seq):
        assert float((a_['raw_dns'][0] - b_['raw_dns'][0]).abs().max()) <= 5e-6
    for a_, b_ in zip(got, seq):
        assert np.allclose(np.asarray(a_['regs'], np.float64), np.asarray(b_['regs'], np.float64), rtol=1e-10, atol=0)
        assert np.allclose(np.asarray(a_['params'], np.float64), np.asarray(b_['params'], np.float64), rtol=1e-10, atol=0)
        assert float((a_['raw_dns'][0] - b_['raw_dns'][0]).abs().max()) <= 5e-6      # two IterDenoise runs differ by up to 1e-6 themselves


@pytest.mark.parametrize("weights", ["denoise", "procedural"])
def test_denoise_stream_iter_matches_iterdenoise(weights):
    """The reference's shipped mode (iter, max_iter 1; YOND_SIDD.py:419-472) on the two-stream driver: frame k's collaborative estimate and
    parameter chain run on the side stream under another frame's network pass (pipeline._denoise_stream_chain_iter), round 2 is queued
    speculatively -- and every frame must come out as IterDenoise returns it, both rounds.  With denoising weights every frame runs two
    rounds; with procedural weights every frame's round 2 ends at the beta1 < 0 guard (:445-447, one output) -- the guard depends on the
    network, so one stream cannot mix the two -- frames of two sizes, more frames than the driver's buffer ring holds."""
    import yond_public_amd.pipeline as P
    import yond_public_amd.archs as A
    import yond_public_amd.synthetic as S
    dev = torch.device('cuda:0')
    arch = dict(name='GuidedResUnet', in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True, guided=True)
    net = A.GuidedResUnet(dict(arch))
    net.load_state_dict(S.denoising_state_dict(net, 0) if weights == "denoise" else S.procedural_state_dict(net, 0))
    net = net.to(dev).eval()
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': True}
    shapes = [(256, 320), (256, 320), (320, 448), (256, 320), (320, 448), (256, 320), (256, 320)]
    frames = [torch.from_numpy(S.synth_noisy(h, w, 3.0 + i, 5.0 + 2 * i, 10 + i)[0]).to(dev) for i, (h, w) in enumerate(shapes)]
    seq = [P.IterDenoise(f, net, arch, pipe) for f in frames]
    assert P.STREAM_ITER
    got = list(P.denoise_stream(iter(frames), net, arch, pipe))
    one = list(P.denoise_stream(iter(frames[:1]), net, arch, pipe))                  # a single frame drains correctly
    # round 6: second passes on a lane of their own (default); every frame's whole chain on one of two lanes (LANE_PIPELINES); one stream (STREAM_LANES 1)
    assert P.STREAM_LANES == 2 and not P.LANE_PIPELINES
    P.LANE_PIPELINES = True
    try:
        got_lp = list(P.denoise_stream(iter(frames), net, arch, pipe))
    finally:
        P.LANE_PIPELINES = False
    P.STREAM_LANES = 1
    try:
        got_1 = list(P.denoise_stream(iter(frames), net, arch, pipe))
    finally:
        P.STREAM_LANES = 2
    torch.cuda.synchronize()
    assert len(got) == len(got_lp) == len(got_1) == len(seq) and len(one) == 1
    want_rounds = 2 if weights == "denoise" else 1
    for a_, b_ in zip(got + one + got_lp + got_1, seq + seq[:1] + seq + seq):
        assert len(a_['raw_dns']) == len(b_['raw_dns']) == want_rounds and len(a_['regs']) == len(b_['regs']) == want_rounds
        assert np.allclose(np.asarray(a_['regs'], np.float64), np.asarray(b_['regs'], np.float64), rtol=1e-9, atol=0)
        assert np.allclose(np.asarray(a_['params'], np.float64), np.asarray(b_['params'], np.float64), rtol=1e-9, atol=0)
        for x, y in zip(a_['raw_dns'], b_['raw_dns']):
            assert float((x - y).abs().max()) <= 5e-6                    # (two IterDenoise runs differ by up to 1e-6 themselves: atomics in the moment sums)


def test_cfg4_unet_batch8_full_size():
    """BASELINE cfg 4 at its real shape: UNetSeeInDark, batch 8 of 3000 x 4000 frames (packed, padded 1504 x 2016) in ONE
    forward; every item equals its batch-1 forward bit for bit (no cross-item coupling, deterministic tile arithmetic)."""
    from yond_public_amd import pipeline as P
    arch = ARCHS["unet32"]
    net, _ = make_net(arch, 2)
    plan = P._plan_of(net, torch.device(DEV))
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.rand((8, 1504, 2016, 4), device=DEV, generator=g) * 0.9
    ub = x.reshape(8, -1).max(dim=1).values.contiguous()
    y = plan.forward_nhwc4(x, None, ub=ub)
    torch.cuda.synchronize()
    assert y.shape == x.shape and bool(torch.isfinite(y).all())
    for i in (0, 3, 7):
        yi = plan.forward_nhwc4(x[i:i + 1].contiguous(), None, ub=ub[i:i + 1].contiguous())
        assert torch.equal(yi[0], y[i]), float((yi[0] - y[i]).abs().max())
    # the batched full-frame driver on two of these sizes' worth of Bayer frames (per-frame estimates, one forward)
    import yond_public_amd.synthetic as S
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'full_dn': True}
    frames = [torch.from_numpy(S.synth_noisy(3000, 4000, 4.0, 6.0, 70 + i)[0]).to(DEV) for i in range(2)]
    bat = P.IterDenoiseBatch(frames + frames, net, arch, pipe)           # batch 4
    one = P.IterDenoise(frames[1], net, arch, pipe)
    assert bat['raw_dns'][0].shape == (4, 3000, 4000)
    assert float((bat['raw_dns'][0][1] - one['raw_dns'][0]).abs().max()) <= 2e-6
    # (the same frame twice: its two estimates differ in the last bits -- float64 atomics -- and so may the outputs)
    assert float((bat['raw_dns'][0][1] - bat['raw_dns'][0][3]).abs().max()) <= 2e-6


def test_cfg5_fp16_path_unclipped_4000x6000():
    """BASELINE cfg 5 at its real shape: one low-light 4000 x 6000 frame without black-level clip (negative DN reach the
    VST), GuidedResUnet with precision='fp16' (fp16 MFMA operands, fp32 accumulate): finite, in range, and >= 55 dB from
    the fp32 path -- on the whole frame and on a 512 x 512 crop."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    arch = ARCHS["gru32"]
    net, _ = make_net(arch, 3)
    H, W, K, sigma = 4000, 6000, 2.0, 25.0
    clean = torch.from_numpy((O.synth_clean(H, W) * 0.2).astype(np.float32)).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(9)
    noisy = (torch.poisson(clean * (959.0 / K), generator=g) * K + torch.randn((H, W), device=DEV, generator=g) * sigma) / 959.0
    noisy = noisy.float().contiguous()
    assert float(noisy.min()) < 0.0                             # the case really has negative pixels
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'full_dn': True}
    r32 = P.IterDenoise(noisy, net, arch, pipe)
    net.precision = 'fp16'
    r16 = P.IterDenoise(noisy, net, arch, pipe)
    net.precision = 'fp32'
    a, b = r16['raw_dns'][0], r32['raw_dns'][0]
    assert a.shape == (H, W) and bool(torch.isfinite(a).all()) and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    np.testing.assert_allclose(r16['regs'][0], r32['regs'][0], rtol=1e-9)         # the estimate does not depend on the conv path
    for name, (u, v) in {"frame": (a, b), "crop": (a[1700:2212, 2700:3212], b[1700:2212, 2700:3212])}.items():
        mse = float(((u.double() - v.double()) ** 2).mean())
        psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
        print(f"[parity] cfg5 4000x6000 fp16-MFMA path vs fp32 path, {name}: PSNR {psnr:.1f} dB")
        assert psnr >= 55.0


# ---- BASELINE.json's configurations at their REAL sizes against outputs of the reference itself (tests/golden/full_cfg*.npz) ----
def _full_net(arch, sd):
    from yond_public_amd import archs as A
    net = getattr(A, arch['name'])(dict(arch))
    net.load_state_dict(sd)
    return net.to(DEV).eval()


def _check_full(name, dn, g, tag, atol=1e-4):
    from test_oracle_golden import full_crops
    dn = dn.cpu().numpy()
    worst = 0.0
    for got, t in zip(full_crops(dn), ("a", "b", "c", "sub")):
        worst = max(worst, report(f"{name} {t}", got, g[f"{tag}_{t}"]))
    chk = g[f"{tag}_chk"]
    s = np.asarray(dn, np.float64)
    np.testing.assert_allclose([s.sum(), np.abs(s).sum(), (s * s).sum()], chk, rtol=2e-6)
    assert worst <= atol
    return dn


def _check_regs(name, r, gr):
    print(f"[parity] {name} regs {r[0]:.8e},{r[1]:.8e} vs reference {gr[0]:.8e},{gr[1]:.8e}")
    np.testing.assert_allclose(r[0], gr[0], rtol=2e-5)
    np.testing.assert_allclose(r[1], gr[1], rtol=0, atol=2e-5 * abs(gr[0]) + 1e-9)


def test_full_cfg2_3000x4000_matches_reference(golden):
    """configs[1] at its real size against THE REFERENCE: the 3000 x 4000 frame bench.py times, GuidedResUnet nf 32 with the
    denoising weights, both rounds.  Round 1 is the reference's IterDenoise as shipped (YOND_SIDD.py:301-410); its round 2 raises at
    this width (np.split of a packed width of 2000 into 32, :91-93), so the fixture's round 2 is :431-458 put together from the
    reference's own SimpleNLF / get_bias / VST_Denoiser without the re-tiling -- what pipeline.IterDenoise does there.  Every tile
    round of the 256 persistent workgroups, the 12-row / 8-row tile choice and the plane offsets of a 12 MP frame are under
    this comparison; PSNR against the clean frame within 0.01 dB of the reference's (BASELINE north_star)."""
    from test_oracle_golden import FULL_PIPE, full_case
    from yond_public_amd import pipeline as P
    g = golden("full_cfg2")
    noisy, clean, arch, sd = full_case("cfg2")
    assert np.array_equal(sha(noisy), g["a_sha"])
    net = _full_net(arch, sd)
    res = P.IterDenoise(noisy, net, arch, dict(FULL_PIPE, iter='iter'), device=DEV)
    assert len(res['raw_dns']) == 2
    _check_regs("cfg2 3000x4000 round 1", res['regs'][0], g["a_reg0"])
    _check_regs("cfg2 3000x4000 round 2", res['regs'][1], g["a_reg1"])
    np.testing.assert_allclose(res['params'][0], g["a_params0"], rtol=2e-5)
    np.testing.assert_allclose(res['params'][1], g["a_params1"], rtol=2e-5)
    for it in range(2):
        dn = _check_full(f"cfg2 3000x4000 round {it + 1} vs reference", res['raw_dns'][it], g, f"a_dn{it}")
        psnr = 10 * np.log10(1.0 / ((np.asarray(dn, np.float64) - clean) ** 2).mean())
        print(f"[parity] cfg2 3000x4000 round {it + 1}: PSNR vs clean {psnr:.5f} dB, reference {g['a_psnr'][it + 1]:.5f} dB")
        assert abs(psnr - g["a_psnr"][it + 1]) <= 0.01
    # 'once' is round 1 alone
    once = P.IterDenoise(noisy, net, arch, dict(FULL_PIPE, iter='once'), device=DEV)
    assert len(once['raw_dns']) == 1
    _check_full("cfg2 3000x4000 once vs reference", once['raw_dns'][0], g, "a_dn0")


def test_full_cfg2_3000x4096_iter_matches_reference(golden):
    """A 12.3 MP frame whose packed width divides by 32: the reference's IterDenoise runs 'iter' AS SHIPPED (SIDD_256 re-tiling of
    the collaborative estimate included, :431) -- both rounds against it."""
    from test_oracle_golden import FULL_PIPE, full_case
    from yond_public_amd import pipeline as P
    g = golden("full_cfg2")
    noisy, clean, arch, sd = full_case("cfg2w")
    assert np.array_equal(sha(noisy), g["b_sha"])
    net = _full_net(arch, sd)
    res = P.IterDenoise(noisy, net, arch, dict(FULL_PIPE, iter='iter'), device=DEV)
    assert len(res['raw_dns']) == int(g["b_nout"]) == 2
    for it in range(2):
        _check_regs(f"3000x4096 round {it + 1}", res['regs'][it], g["b_regs"][it])
        _check_full(f"3000x4096 round {it + 1} vs reference", res['raw_dns'][it], g, f"b_dn{it}")


def test_full_cfg4_unet_two_frames_match_reference(golden):
    """configs[3]'s shape against the reference: UNetSeeInDark on two 3000 x 4000 frames -- the reference denoises them one by one
    (batch 1), IterDenoiseBatch in ONE batched forward with per-frame estimates."""
    from test_oracle_golden import FULL_PIPE, full_case
    from yond_public_amd import pipeline as P
    g = golden("full_cfg4")
    cases = [full_case("cfg4", i) for i in range(2)]
    for i in range(2):
        assert np.array_equal(sha(cases[i][0]), g[f"sha_{i}"])
    arch, sd = cases[0][2], cases[0][3]
    net = _full_net(arch, sd)
    frames = [torch.from_numpy(c[0]).to(DEV) for c in cases]
    bat = P.IterDenoiseBatch(frames, net, arch, dict(FULL_PIPE, iter='once'))
    assert bat['raw_dns'][0].shape == (2, 3000, 4000)
    for i in range(2):
        _check_regs(f"cfg4 frame {i}", bat['regs'][0][i], g[f"reg_{i}"])
        _check_full(f"cfg4 UNetSeeInDark frame {i} vs reference", bat['raw_dns'][0][i], g, f"dn_{i}")


def test_full_cfg5_4000x6000_unclipped_matches_reference(golden):
    """configs[4]'s input side against the reference: a 4000 x 6000 low-light frame without black-level clip (negative DN reach
    the VST; the reference's functions composed as :341, :356, :384-389 -- its own np.split at :354 needs a width that divides by 32).
    The float32-accurate path to 1e-4; the fp16-MFMA path (precision='fp16', what cfg 5 times) >= 55 dB from the REFERENCE's output."""
    from test_oracle_golden import FULL_PIPE, full_case, full_crops
    from yond_public_amd import pipeline as P
    g = golden("full_cfg5")
    noisy, clean, arch, sd = full_case("cfg5")
    assert np.array_equal(sha(noisy), g["sha"]) and float(noisy.min()) < 0
    net = _full_net(arch, sd)
    pipe = dict(FULL_PIPE, iter='once')
    res = P.IterDenoise(noisy, net, arch, pipe, device=DEV)
    _check_regs("cfg5 4000x6000", res['regs'][0], g["reg"])
    np.testing.assert_allclose(res['params'][0], g["params"], rtol=2e-5)
    _check_full("cfg5 4000x6000 fp32 path vs reference", res['raw_dns'][0], g, "dn")
    net.precision = 'fp16'
    r16 = P.IterDenoise(noisy, net, arch, pipe, device=DEV)
    net.precision = 'fp32'
    got = np.concatenate([c.reshape(-1) for c in full_crops(r16['raw_dns'][0].cpu().numpy())]).astype(np.float64)
    ref = np.concatenate([g[f"dn_{t}"].reshape(-1) for t in ("a", "b", "c", "sub")]).astype(np.float64)
    psnr = 10 * np.log10(1.0 / max(((got - ref) ** 2).mean(), 1e-30))
    print(f"[parity] cfg5 4000x6000 fp16-MFMA path vs the reference's output: PSNR {psnr:.1f} dB")
    assert psnr >= 55.0
    # ... and the fp16 path denoises as well as the float32 path: PSNR against the CLEAN frame within 0.01 dB (north_star)
    cl = np.asarray(clean, np.float64)
    pc = lambda a: 10 * np.log10(1.0 / float(np.mean((np.clip(a.cpu().numpy().astype(np.float64), 0, 1) - np.clip(cl, 0, 1)) ** 2)))
    p32, p16 = pc(res['raw_dns'][0]), pc(r16['raw_dns'][0])
    print(f"[parity] cfg5 4000x6000 PSNR vs clean: fp32 path {p32:.4f} dB, fp16 path {p16:.4f} dB")
    assert abs(p32 - p16) <= 0.01
