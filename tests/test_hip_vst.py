"""GPU: K1 / K4 / pack / bias LUT against the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from hip_common import report

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_pack_unpack_bit_exact(golden):
    from yond_public_amd import pipeline as P
    g = golden("pack")
    assert np.array_equal(P.bayer2rggb(g["bayer"], DEV).cpu().numpy(), g["rggb"])
    assert np.array_equal(P.rggb2bayer(g["rggb_in"], DEV).cpu().numpy(), g["bayer_out"])
    a = np.random.default_rng(0).random((3000, 4000)).astype(np.float32)       # BASELINE cfg-2 size: round trip
    t = torch.from_numpy(a).to(DEV)
    assert torch.equal(P.rggb2bayer(P.bayer2rggb(t)), t)


def test_bias_lut_matches_reference_knots(golden):
    from yond_public_amd import pipeline as P
    g = golden("bias")
    worst = 0.0
    for i, (K, s) in enumerate(g["ksig"]):
        for tag, mx in zip("abc", g["max"]):
            lut = P.get_bias(np.float32(mx), np.float64(s), np.float64(K), device=DEV)
            np.testing.assert_array_equal(lut.lams, g[f"lams_{i}{tag}"])
            got = lut.y.cpu().numpy()
            ref = g[f"bias_{i}{tag}"]
            worst = max(worst, report(f"bias LUT K={K} s={s} max={mx}", got, ref))
            # float32 knots: a few ulp of float32 at |bias| <= 0.2 (different exp/lgamma/summation order)
            np.testing.assert_allclose(got, ref, rtol=0, atol=3e-7)
    print("worst bias knot error", worst)


@pytest.mark.parametrize("H,W,K,s,bc", [(128, 192, 4.37, 6.27, True), (120, 136, 22.65, 37.09, True), (64, 96, 0.72, 1.8, True),
                                        (128, 128, 4.37, 6.27, False)])
def test_k1_pack_vst_norm(H, W, K, s, bc):
    import yond_oracle as O
    import torch.nn.functional as F
    from yond_public_amd import pipeline as P, _lib as L
    lib = L.load()
    noisy, _ = O.synth_noisy(H, W, K, s, 3)
    # pixels that land EXACTLY on the LUT's knots, the repeated ones (50, 500: get_bias concatenates its runs) included
    hits = []
    for target in (50.0, 500.0, 49.9, 0.1, 51.0, 510.0):
        q = np.float32(target / 959.0)
        for _ in range(8):
            if np.float32(q * np.float32(959.0)) == np.float32(target):
                hits.append(q)
                break
            q = np.nextafter(q, np.float32(2.0 if np.float32(q * np.float32(959.0)) < target else 0.0), dtype=np.float32)
    assert len(hits) >= 3
    noisy.reshape(-1)[7:7 + len(hits)] = hits
    K, s = np.float64(K), np.float64(s)
    # oracle staging (YOND_SIDD.py:251-269, 281-286)
    lr = O.bayer2rggb(noisy) * 959.0
    v = O.VST(lr, s, gain=K)
    if bc:
        f = O.get_bias(lr.max(), s, K)
        v = v - f(np.maximum(lr, 0))
    lo, hi = O.VST(0, s, gain=K), O.VST(959.0, s, gain=K)
    u = torch.from_numpy(np.ascontiguousarray((v - lo) / (hi - lo))).float().permute(2, 0, 1)[None]
    p2d = O.get_p2d(u.shape, 32)
    ref = F.pad(u, p2d, mode='reflect').clamp(0, 1)[0].permute(1, 2, 0).numpy()
    # HIP
    t = torch.from_numpy(noisy).to(DEV)
    h, w = H // 2, W // 2
    Hp, Wp = h + p2d[2] + p2d[3], w + p2d[0] + p2d[1]
    x4 = torch.empty((Hp, Wp, 4), device=DEV)
    mx = torch.empty(1, device=DEV)
    lut = P.get_bias(lr.max(), s, K, device=DEV) if bc else None
    L.check(lib.yond_pack_vst_norm_f32(L.ptr(t), H, W, L.ptr(x4), *p2d, 1, 959.0, float(K), float(s), float(lo), float(hi),
                                       L.ptr(lut.x) if bc else None, L.ptr(lut.y) if bc else None, len(lut) if bc else 0,
                                       L.ptr(mx), L.stream()), "k1")
    got = x4.cpu().numpy()
    err = report(f"K1 {H}x{W} K={K}", got, ref)
    assert err <= 3e-7          # <= 2-3 ulp of float32 on [0,1] (LUT knots differ by float32 ulps)
    assert float(mx.item()) == float(got.max())


def test_k4_denorm_ivst_unpack():
    import yond_oracle as O
    from yond_public_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(5)
    Hp, Wp, h, w, pt, pl = 64, 96, 60, 90, 2, 3
    y = (rng.random((Hp, Wp, 4)) * 1.2 - 0.1).astype(np.float32)
    K, s = np.float64(4.37), np.float64(6.27)
    lo, hi = O.VST(0, s, gain=K), O.VST(959.0, s, gain=K)
    for mode, exact in ((1, False), (2, True)):
        yc = np.clip(y, 0, 1)[pt:pt + h, pl:pl + w]
        z = yc * (hi - lo) + lo
        ref = O.rggb2bayer(O.inverse_VST(z, s, gain=K, exact=exact)) / 959.0
        out = torch.empty((2 * h, 2 * w), device=DEV)
        yd = torch.from_numpy(y).to(DEV)
        L.check(lib.yond_denorm_ivst_unpack_f32(L.ptr(yd), Hp, Wp, pt, pl, h, w, L.ptr(out), mode, 959.0,
                                                float(K), float(s), float(lo), float(hi), 0, L.stream()), "k4")
        got = out.cpu().numpy()
        err = np.abs(got.astype(np.float64) - ref)
        rel = err / np.maximum(np.abs(ref), 1e-6)
        print(f"[parity] K4 mode {mode}: max_abs={err.max():.3e} max_rel={rel.max():.3e}")
        assert rel.max() <= 2 ** -23        # one rounding to float32


@pytest.mark.parametrize("ci", range(4))
def test_vst_denoiser_matches_reference(golden, ci):
    """End-to-end row J against the reference's own output (tests/golden/vst_denoiser.npz)."""
    import yond_oracle as O
    from hip_common import ARCHS, make_net, sha
    from yond_public_amd import pipeline as P
    g = golden("vst_denoiser")
    H, W, K, s, idx, seed = g[f"meta_{ci}"]
    arch = ARCHS[str(g[f"arch_{ci}"])]
    bc = str(g[f"bias_corr_{ci}"])
    bc = None if bc == 'None' else bc
    noisy, _ = O.synth_noisy(int(H), int(W), K, s, int(idx))
    assert np.array_equal(sha(noisy), g[f"sha_{ci}"])
    net, sd = make_net(arch, int(seed))
    p = {'wp': 1023, 'bl': 64, 'ratio': 1, 'scale': 959.0, 'gain': np.float64(K), 'sigma': np.float64(s)}
    dn = P.VST_Denoiser(torch.from_numpy(noisy).to(DEV), p, net, arch, bias_corr=bc).cpu().numpy()
    ref = g[f"dn_{ci}"]
    err = report(f"VST_Denoiser case {ci}", dn, ref)
    assert err <= 1e-4


def test_bias_lut_2d_matches_reference(golden):
    """Row H' (utils/isp_algos.py:162-231): the 2-D bias LUT lookup on the device -- in-table interpolation, the constant /
    closed-form region beyond the table, the sigma-outside fallback to get_bias -- against the reference's own
    BiasLUT.get_lut on a small table built with the reference's get_bias_points (tests/golden/biaslut.npz)."""
    from yond_public_amd import pipeline as P
    g = golden("biaslut")
    lut = P.BiasLUT(table=g["table"], x_lut=g["x_lut"], sg_lut=g["sg_lut"])
    for ci in range(int(g["ncases"])):
        K, s = (np.float64(v) for v in g[f"ksig_{ci}"])
        x = g[f"x_{ci}"]
        got = lut.get_lut(torch.from_numpy(x).to(DEV), K=K, sigGs=s).cpu().numpy()
        ref = g[f"bias_{ci}"]
        err = report(f"BiasLUT.get_lut K={K} sigma={s}", got, ref)
        # in-table: float64 interpolation (1e-12); sigma outside the table (last case): the device-built 1-D LUT, float32 knots
        assert err <= (3e-7 if lut.row(K, s, DEV) is None else 1e-10)


def test_vst_denoiser_with_bias_lut_2d(golden):
    """YOND_SIDD.py:254-259 with self.biaslut set: K1 evaluates the merged table row per pixel."""
    import yond_oracle as O
    from hip_common import ARCHS, make_net, sha
    from yond_public_amd import pipeline as P
    g = golden("biaslut")
    lut = P.BiasLUT(table=g["table"], x_lut=g["x_lut"], sg_lut=g["sg_lut"])
    noisy, _ = O.synth_noisy(96, 128, 4.37, 6.27, 55)
    assert np.array_equal(sha(noisy), g["sha_vd"])
    arch = ARCHS["gru8"]
    net, sd = make_net(arch, 91)
    p = {'wp': 1023, 'bl': 64, 'ratio': 1, 'scale': 959.0, 'gain': np.float64(4.37), 'sigma': np.float64(6.27)}
    dn = P.VST_Denoiser(torch.from_numpy(noisy).to(DEV), p, net, arch, bias_corr='pre', biaslut=lut).cpu().numpy()
    assert report("VST_Denoiser through the 2-D bias LUT", dn, g["dn_vd"]) <= 1e-4
    # and the estimator-driven pipeline accepts the table (full frame and SIDD stack)
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'full_dn': True}
    ref = O.IterDenoise(noisy, arch, sd, pipe)      # 1-D LUT path of the oracle: the two LUTs agree to the table's resolution
    res = P.IterDenoise(noisy, net, arch, pipe, device=DEV, biaslut=lut)
    assert float(np.abs(res['raw_dns'][0].cpu().numpy() - ref['raw_dns'][0]).max()) < 5e-3


def test_bias_eval_on_drifting_and_float32_grids():
    """K1's interval lookup trusts evenly spaced runs (one multiply instead of a search).  A slowly drifting grid (log spacing of
    ratio 1.0005: neighbouring intervals differ by less than the break tolerance) must fall back to the bisection, not be
    extrapolated from the wrong interval; a grid whose knots are float32-rounded (what NumPy 2 produces for a float32 maximum)
    must still take the fast path and agree with interp1d."""
    from scipy.interpolate import interp1d
    from yond_public_amd import pipeline as P
    rng = np.random.default_rng(5)
    for name, x in (("log 1.0005", 10.0 * 1.0005 ** np.arange(1500)),
                    ("float32 linspace", np.concatenate((np.linspace(0, 50, 501), np.linspace(50, 500, 451),
                                                         np.linspace(np.float32(500), np.float32(961), 48).astype(np.float64))))):
        y = np.sin(x / 40.0).astype(np.float32)
        f = P.DeviceBiasLUT(x, torch.from_numpy(np.ascontiguousarray(x, np.float64)).to(DEV), torch.from_numpy(y).to(DEV))
        xq = rng.uniform(x[0], x[-1], 20000).astype(np.float32)
        xq[:len(x)] = x.astype(np.float32).clip(x[0], x[-1])          # the knots themselves
        got = f(xq).cpu().numpy()
        want = interp1d(x, y.astype(np.float64))(xq.astype(np.float64).clip(x[0], x[-1]))
        err = np.abs(got - want).max()
        print(f"[parity] bias LUT on a {name} grid: max |delta| = {err:.3e}")
        assert err < 1e-7


def test_bias_lut_large_gain_sigma_fallback():
    """14-bit frames at a digital gain (ELD at ratio 10: K ~ 80, sigma ~ 300 DN): the Gaussian table of the integration
    (~16 k float64 per knot) exceeds the LDS; get_bias then runs the same kernel with the table in a global scratch buffer.
    Knots and ordinates against the oracle's restatement of utils/isp_algos.py:98-140."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    from yond_public_amd import _lib as L
    K, s, mx = np.float64(80.0), np.float64(300.0), np.float32(140.3)
    assert L.load().yond_bias_lut_f64(None, 1, float(K), float(s), None, None) != 0          # (arguments rejected before any launch)
    lut = P.get_bias(mx, s, K, device=DEV)
    ref = O.get_bias(mx, s, K)
    np.testing.assert_array_equal(lut.lams, np.asarray(ref.x))
    got, want = lut.y.cpu().numpy(), np.asarray(ref.y)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
    print(f"[parity] large-table bias LUT (K=80, sigma=300): {len(got)} knots, max |delta| = {err:.3e}, |bias| <= {np.abs(want).max():.3e}")
    assert err <= 2e-7 * max(1.0, float(np.abs(want).max()))


def test_get_bias_points_and_biaslut_leftovers(golden):
    """utils/isp_algos.py:142-160 get_bias_points (pho_min = 100) on the device against the oracle's restatement, and the two
    branches of BiasLUT.get_lut that use it / were missing: <= 1000 points with sigma outside the table (:211-212) and
    func=True (:205-206, :216-219)."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    g = golden("biaslut")
    x_lut, sg_lut, table = g["x_lut"], g["sg_lut"], g["table"]
    rng = np.random.default_rng(2)
    lams = np.concatenate(([0.0, 0.3, 5.0], rng.uniform(0, 60, 9)))
    for (K, s, cf) in ((2.0, 3.0, False), (4.0, 45.0, True)):
        got = P.get_bias_points(lams, K, s, pho_min=100, close_form=cf, device=DEV).cpu().numpy()
        want = O.get_bias_points(lams.copy(), K, s, pho_min=100, close_form=cf)
        err = np.abs(got - want).max()
        print(f"[parity] get_bias_points K={K} sigma={s} close_form={cf}: max |delta| = {err:.3e}")
        assert err < 1e-10
    lut_d = P.BiasLUT(table=table, x_lut=x_lut, sg_lut=sg_lut)
    lut_o = O.BiasLUT(table, x_lut, sg_lut)
    K, s = np.float64(2.0), np.float64(2.0 * sg_lut[-1] * 1.5)                   # sigma / K beyond the sigma grid
    xq = rng.uniform(0, 80, (5, 7)).astype(np.float32)                           # 35 points: the pointwise branch
    got = lut_d.get_lut(torch.from_numpy(xq).to(DEV), K=K, sigGs=s).cpu().numpy()
    want = lut_o.get_lut(xq.copy(), K=K, sigGs=s)
    assert got.shape == want.shape and np.abs(got.astype(np.float32) - want).max() < 1e-6     # (the reference keeps the queries' float32)
    f_out = lut_d.get_lut(torch.from_numpy(xq).to(DEV), K=K, sigGs=s, func=True)              # outside: get_bias' interp1d object
    f_ref = lut_o.get_lut(xq.copy(), K=K, sigGs=s, func=True)
    np.testing.assert_array_equal(np.asarray(f_out.lams, np.float64), np.asarray(f_ref.x, np.float64))
    assert np.abs(f_out(xq).cpu().numpy() - f_ref(xq)).max() < 1e-6
    K2, s2 = np.float64(2.0), np.float64(2.0 * sg_lut[5])                        # inside: the row as a callable
    f_in = lut_d.get_lut(torch.from_numpy(xq).to(DEV), K=K2, sigGs=s2, func=True)
    np.testing.assert_allclose(f_in(xq).cpu().numpy(), lut_o.get_lut(xq.astype(np.float64), K=K2, sigGs=s2), rtol=1e-12, atol=1e-12)


def test_manual_est_type_and_refused_est_types():
    """YOND_SIDD.py:349-351: est_type 'manual' runs round 1 at (K, sigma) = (14, 20) DN; the est_type that needs a second network is
    refused loudly, and so is a file est_type without the directory to read from."""
    import yond_oracle as O
    from hip_common import ARCHS
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    from yond_public_amd import synthetic as S
    arch = ARCHS["gru8"]
    net = A.GuidedResUnet(dict(arch))
    net.load_state_dict(S.denoising_state_dict(net, 3))
    net = net.to(DEV).eval()
    noisy, _ = O.synth_noisy(256, 384, 14.0, 20.0, 5)
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'max_iter': 1, 'full_dn': True, 'est_type': 'manual'}
    res = P.IterDenoise(noisy, net, arch, pipe, device=DEV)
    assert abs(res['params'][0][0] - 14.0) < 1e-12 and abs(res['params'][0][1] - 20.0) < 1e-12
    torch.set_num_threads(8)
    ref = O.IterDenoise(noisy, arch, S.denoising_state_dict(net, 3), pipe)
    assert float(np.abs(res['raw_dns'][0].cpu().numpy() - ref['raw_dns'][0]).max()) <= 1e-5
    with pytest.raises(NotImplementedError):                    # a second network (NeuralNLF): not built, refused loudly
        P.IterDenoise(noisy, net, arch, dict(pipe, est_type='ours'), device=DEV)
    from yond_public_amd._lib import YondHipError
    with pytest.raises(YondHipError):                           # a file est_type without the dataset directory to read from
        P.IterDenoise(noisy, net, arch, dict(pipe, est_type='foi'), device=DEV)


def test_file_lookup_est_types_run_round_one_at_the_stored_estimate(tmp_path):
    """YOND_SIDD.py:316-337: est_types 'foi' / 'liu' / 'zou' / 'pge' and pipe['cal_est'] take round 1's (beta1, beta2) from files.  With
    the files holding 'manual''s fixed estimate (14, 20 DN) every one of them must reproduce the 'manual' run bit for bit; round 2 of
    'iter' then estimates collaboratively as usual."""
    import pickle
    import scipy.io as sio
    import yond_oracle as O
    from hip_common import ARCHS
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    from yond_public_amd import synthetic as S
    arch = ARCHS["gru8"]
    net = A.GuidedResUnet(dict(arch))
    net.load_state_dict(S.denoising_state_dict(net, 3))
    net = net.to(DEV).eval()
    noisy, _ = O.synth_noisy(256, 384, 14.0, 20.0, 5)
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': True, 'est_type': 'manual'}
    want = P.IterDenoise(noisy, net, arch, pipe, device=DEV)
    scale = 1023.0 - 64.0
    b1, b2 = 14 / scale, (20 / scale) ** 2
    raw = tmp_path / 'SIDD_Validation_Raw'
    raw.mkdir()
    table = np.zeros((5, 2))
    table[3] = (b1, b2)
    sio.savemat(str(raw / 'FoiEst_fullPict.mat'), {'return_params': table})
    sio.savemat(str(raw / 'LiuEst_fullPict.mat'), {'return_params': table})
    np.save(str(raw / 'Zou_fullPict.npy'), table)
    pge = table.copy()
    pge[3, 1] = np.sqrt(b2)                                     # (PGE stores a standard deviation, :337)
    np.save(str(raw / 'PGE_fullPict.npy'), pge)
    cal = tmp_path / 'cal.pkl'
    with open(cal, 'wb') as f:
        pickle.dump({'sfrn': {'GP_00800': (b1, b2)}, 'beta1': {'GP': [0.0, b1]}, 'beta2': {'GP': [0.0, b2]}}, f)
    est = {'root_dir': str(tmp_path), 'img_id': 3, 'name': '0001_001_GP_00800_00350_3200_N'}
    runs = [dict(pipe, est_type=t) for t in ('foi', 'liu', 'zou', 'pge')] + [dict(pipe, est_type='simple', cal_est=str(cal))]
    ests = [est] * 4 + [est, dict(est, name='0001_001_GP_00100_00350_3200_N')]      # (the last: an ISO without a record -> the polynomials)
    runs.append(dict(pipe, est_type='simple', cal_est=str(cal)))
    for pp, e in zip(runs, ests):
        got = P.IterDenoise(noisy, net, arch, pp, device=DEV, est=e)
        assert len(got['raw_dns']) == len(want['raw_dns']) == 2
        np.testing.assert_allclose(got['regs'][0], want['regs'][0], rtol=1e-12)
        for a_, b_ in zip(got['raw_dns'], want['raw_dns']):
            assert float((a_ - b_).abs().max()) <= 1e-6, pp


def test_batched_k1_k4_equal_the_per_frame_kernels():
    """yond_pack_vst_norm_batch_f32 / yond_denorm_ivst_unpack_batch_f32 (one launch for the 32 blocks of a SIDD image) against 32
    calls of the per-frame kernels: every bit of the packed tensor, of the per-block maxima and of the unpacked frames."""
    import yond_oracle as O
    from yond_public_amd import _lib as L, pipeline as P
    lib = L.load()
    B, H, W = 5, 64, 96
    rng = np.random.default_rng(4)
    lr = torch.from_numpy(rng.random((B, H, W), dtype=np.float32)).to(DEV)
    K, sig, scale = np.float64(4.37), np.float64(6.27), 959.0
    f = P.get_bias(np.float32(lr.max().item()) * np.float32(scale), sig, K, device=DEV)
    lower, upper = P.vst_scalar(0, sig, K), P.vst_scalar(scale, sig, K)
    p2d = P.get_p2d((1, 4, H // 2, W // 2), base=32)
    Hp, Wp = H // 2 + p2d[2] + p2d[3], W // 2 + p2d[0] + p2d[1]
    st = L.stream()
    xa, ma = torch.zeros(B, Hp, Wp, 4, device=DEV), torch.zeros(B, device=DEV)
    xb, mb = torch.ones(B, Hp, Wp, 4, device=DEV), torch.ones(B, device=DEV)
    for i in range(B):
        L.check(lib.yond_pack_vst_norm_f32(L.ptr(lr[i]), H, W, L.ptr(xa[i]), p2d[0], p2d[1], p2d[2], p2d[3], 1, scale, float(K), float(sig),
                                           float(lower), float(upper), L.ptr(f.x), L.ptr(f.y), len(f), L.ptr(ma[i:i + 1]), st), "k1")
    L.check(lib.yond_pack_vst_norm_batch_f32(L.ptr(lr), B, H, W, L.ptr(xb), p2d[0], p2d[1], p2d[2], p2d[3], scale, float(K), float(sig),
                                             float(lower), float(upper), L.ptr(f.x), L.ptr(f.y), len(f), 0, L.ptr(mb), st), "k1 batch")
    torch.cuda.synchronize()
    assert torch.equal(xa, xb) and torch.equal(ma, mb)
    oa, ob = torch.zeros(B, H, W, device=DEV), torch.ones(B, H, W, device=DEV)
    y = torch.from_numpy(rng.random((B, Hp, Wp, 4), dtype=np.float32)).to(DEV)
    for i in range(B):
        L.check(lib.yond_denorm_ivst_unpack_f32(L.ptr(y[i]), Hp, Wp, p2d[2], p2d[0], H // 2, W // 2, L.ptr(oa[i]), 1, scale, float(K), float(sig),
                                                float(lower), float(upper), 1, st), "k4")
    L.check(lib.yond_denorm_ivst_unpack_batch_f32(L.ptr(y), B, Hp, Wp, p2d[2], p2d[0], H // 2, W // 2, L.ptr(ob), 1, scale, float(K), float(sig),
                                                  float(lower), float(upper), 1, st), "k4 batch")
    torch.cuda.synchronize()
    assert torch.equal(oa, ob)
