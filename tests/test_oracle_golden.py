"""CPU: the oracle (oracle/yond_oracle.py) replayed against golden vectors produced by running
the reference itself (oracle/gen_golden.py).  These pin the oracle; see its header for the one
unpinned boundary (cv2.blur)."""
import hashlib
import os

import numpy as np
import pytest
import torch

import yond_oracle as O

ARCHS = {
    "gru32": dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
    "gru8": dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=8, nframes=1, res=True, norm=True),
    "gru32_nonorm": dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=False, norm=False),
    "snr32": dict(name='SNRnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
    "snr8": dict(name='SNRnet', guided=True, in_nc=4, out_nc=4, nf=8, nframes=1, res=True, norm=True),
    "unet32": dict(name='UNetSeeInDark', in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
    "unet8": dict(name='UNetSeeInDark', in_nc=4, out_nc=4, nf=8, nframes=1, res=True, norm=True),
}


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def test_pack_unpack_bit_exact(golden):
    g = golden("pack")
    assert np.array_equal(O.bayer2rggb(g["bayer"]), g["rggb"])
    assert np.array_equal(O.rggb2bayer(g["rggb_in"]), g["bayer_out"])
    a = np.random.default_rng(0).random((10, 14)).astype(np.float32)
    assert np.array_equal(O.rggb2bayer(O.bayer2rggb(a)), a)


def test_vst_and_inverse(golden):
    g = golden("vst")
    x = g["x"]
    for i, (K, s) in enumerate(g["ksig"]):
        K, s = np.float64(K), np.float64(s)
        v = O.VST(x, s, gain=K)
        assert v.dtype == np.float64
        np.testing.assert_array_equal(v, g[f"vst_{i}"])
        np.testing.assert_array_equal(O.inverse_VST(v, s, gain=K), g[f"ivst_{i}"])
        np.testing.assert_allclose(O.inverse_VST(v, s, gain=K, exact=True), g[f"ivst_exact_{i}"], rtol=1e-14, atol=0)
        z = g[f"z_{i}"]
        np.testing.assert_array_equal(O.inverse_VST(z, s, gain=K), g[f"ivst_z_{i}"])
        np.testing.assert_allclose(O.inverse_VST(z, s, gain=K, exact=True), g[f"ivst_exact_z_{i}"], rtol=1e-14, atol=0)


def test_bias_lut_knots_and_interp(golden):
    g = golden("bias")
    for i, (K, s) in enumerate(g["ksig"]):
        for tag, mx in zip("abc", g["max"]):
            lams, bias = O.get_bias_table(np.float32(mx), np.float64(s), np.float64(K))
            np.testing.assert_array_equal(lams, g[f"lams_{i}{tag}"])
            np.testing.assert_array_equal(bias, g[f"bias_{i}{tag}"])
            f = O.BiasFunc(lams, bias)
            np.testing.assert_array_equal(f(g[f"xq_{i}{tag}"]), g[f"bq_{i}{tag}"])
    with pytest.raises(ValueError):
        O.BiasFunc(lams, bias)(np.array([lams[-1] + 1.0]))


def test_box_blur_two_restatements_agree():
    rng = np.random.default_rng(3)
    a = rng.random((40, 52, 4)).astype(np.float32)
    for k in (5, 19, 29):
        b1, b2 = O.box_blur(a, k), O.box_blur_direct(a, k)
        assert b1.dtype == np.float32
        # the two summation orders may differ in the last float32 bit
        assert np.max(np.abs(b1.astype(np.float64) - b2)) <= 6e-8


def test_percentile_matches_numpy():
    rng = np.random.default_rng(5)
    for n in (7, 1000, 4097):
        a = rng.random(n).astype(np.float32) ** 2
        q = np.linspace(5, 100, 20)
        np.testing.assert_array_equal(O.percentile_linear(a, q), np.percentile(a, q, method='linear'))
    np.testing.assert_array_equal(O.percentile_linear(a, [25.0]), np.percentile(a, [25.0], method='linear'))


@pytest.mark.parametrize("tag", ["s256", "s512", "hi", "lo"])
def test_nle_self_and_collab(golden, tag):
    g = golden("nle")
    H, W, K, s, idx = g[f"{tag}_meta"]
    noisy, clean = O.synth_noisy(int(H), int(W), K, s, int(idx))
    assert np.array_equal(sha(noisy), g[f"{tag}_sha"]), "synthetic input is not reproducible on this box"
    rggb = O.bayer2rggb(noisy)
    mean = O.box_blur(rggb, 29)
    std = O.stdfilt(rggb, 29)
    lap = O.stdfilt(O.box_blur(rggb, 19), 29)
    # float32 maps: the two cv2.blur restatements (cumsum vs ndimage) may differ in the last bit
    np.testing.assert_allclose(mean[:48, :48], g[f"{tag}_mean_crop"], rtol=2e-7, atol=0)
    np.testing.assert_allclose(std[:48, :48], g[f"{tag}_std_crop"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(lap[:48, :48], g[f"{tag}_lap_crop"], rtol=0, atol=2e-6)
    reg, info = O.SimpleNLF(noisy, k=29, setting={'mode': 'self'}, full=True)
    th, pct, b1, b2 = g[f"{tag}_self"]
    assert info["percent"] == pct
    np.testing.assert_allclose(info["th"], th, rtol=1e-5)
    np.testing.assert_allclose(reg[0], b1, rtol=1e-5)
    np.testing.assert_allclose(reg[1], b2, rtol=0, atol=1e-5 * abs(b1) + 1e-9)
    # moment-sum fit == lstsq fit
    dn = np.clip(clean + 0.002 * np.sin(np.arange(int(W))[None, :] / 37.0), 0, 1).astype(np.float32)
    regc, infoc = O.SimpleNLF(noisy, dn, k=29, setting={'mode': 'collab'}, full=True)
    thc, pctc, c1, c2 = g[f"{tag}_collab"]
    assert infoc["percent"] == pctc
    np.testing.assert_allclose(infoc["th"], thc, rtol=1e-5)
    np.testing.assert_allclose(regc[0], c1, rtol=1e-5)
    np.testing.assert_allclose(regc[1], c2, rtol=0, atol=1e-5 * abs(c1) + 1e-9)


def test_nle_sidd256_retiling(golden):
    g = golden("nle")
    noisy, clean = O.synth_noisy(256, 8192, 4.0, 6.0, 7)
    assert np.array_equal(sha(noisy), g["strip_sha"])
    dn = np.clip(clean + 0.002 * np.sin(np.arange(8192)[None, :] / 37.0), 0, 1).astype(np.float32)
    r1 = O.SimpleNLF(noisy, k=29, setting={'mode': 'self', 'SIDD_256': True})
    r2 = O.SimpleNLF(noisy, dn, k=29, setting={'mode': 'collab', 'SIDD_256': True})
    r3 = O.SimpleNLF(noisy, k=29, setting={'mode': 'self'})
    for r, gr in zip((r1, r2, r3), g["strip_regs"]):
        np.testing.assert_allclose(r[0], gr[0], rtol=1e-5)
        np.testing.assert_allclose(r[1], gr[1], rtol=0, atol=1e-5 * abs(gr[0]) + 1e-9)


def test_polyfit_moments_equals_lstsq():
    rng = np.random.default_rng(11)
    m = rng.random(20000).astype(np.float32)
    v = (0.004 * m + 4e-5 + 1e-5 * rng.standard_normal(20000)).astype(np.float32)
    a, b = O.polyfit(m, v), O.polyfit_moments(m, v)
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("ci", range(7))
def test_net_forward(golden, ci):
    g = golden("net")
    aname = str(g[f"arch_{ci}"])
    arch = ARCHS[aname]
    meta = g[f"meta_{ci}"]
    shape = tuple(int(v) for v in meta[2:])
    sd = O.procedural_state_dict(arch, seed=int(meta[0]))
    assert sum(v.numel() for v in sd.values()) == int(g[f"nparams_{ci}"])
    x = torch.rand(shape, generator=torch.Generator().manual_seed(int(meta[1]))) * 0.9
    t = torch.from_numpy(g[f"t_{ci}"]) if f"t_{ci}" in g.files else None
    torch.set_num_threads(8)
    y = O.net_forward(arch, sd, x, t).numpy()
    ref = g[f"y_{ci}"]
    assert y.shape == ref.shape
    # same ATen CPU kernels, functional vs nn.Module call path
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6)


def test_gru32_param_count():
    sd = O.procedural_state_dict(ARCHS["gru32"])
    assert sum(v.numel() for v in sd.values()) == 11173668          # logs/...log:3-4: 11.17 M
    assert tuple(sd['conv1.gamma.0.weight'].shape) == (32, 1, 1, 1)
    assert tuple(sd['upv6.weight'].shape) == (512, 256, 2, 2)
    assert tuple(sd['conv6.short_cut.0.weight'].shape) == (256, 512, 1, 1)


@pytest.mark.parametrize("ci", range(4))
def test_vst_denoiser(golden, ci):
    g = golden("vst_denoiser")
    H, W, K, s, idx, seed = g[f"meta_{ci}"]
    arch = ARCHS[str(g[f"arch_{ci}"])]
    bc = str(g[f"bias_corr_{ci}"])
    bc = None if bc == 'None' else bc
    noisy, _ = O.synth_noisy(int(H), int(W), K, s, int(idx))
    assert np.array_equal(sha(noisy), g[f"sha_{ci}"])
    sd = O.procedural_state_dict(arch, int(seed))
    p = {'wp': 1023, 'bl': 64, 'ratio': 1, 'scale': 959.0, 'gain': np.float64(K), 'sigma': np.float64(s)}
    torch.set_num_threads(8)
    dn = O.VST_Denoiser(noisy, p, arch, sd, bias_corr=bc)
    ref = g[f"dn_{ci}"]
    assert dn.dtype == ref.dtype and dn.shape == ref.shape
    np.testing.assert_allclose(dn, ref, rtol=0, atol=5e-6)


def iter_case(g, ci):
    """Inputs of IterDenoise fixture case `ci` (oracle/gen_golden.py ITER_CASES), regenerated from seeds."""
    K, s = (float(v) for v in g[f"ksig_{ci}"])
    noisy, clean = O.synth_noisy(256, 8192, K, s, 31)
    full, _ = O.synth_noisy(512, 1024, K, s, 32)
    assert np.array_equal(sha(noisy), g[f"sha_noisy_{ci}"]) and np.array_equal(sha(full), g[f"sha_full_{ci}"])
    arch = ARCHS[str(g[f"arch_{ci}"])]
    seed = int(g[f"seed_{ci}"])
    sd = O.denoising_state_dict(arch, seed) if str(g[f"weights_{ci}"]) == "denoise" else O.procedural_state_dict(arch, seed)
    full_dn = bool(g[f"full_dn_{ci}"])
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': full_dn}
    return np.array(np.split(noisy, 32, axis=-1)), clean, full, arch, sd, pipe     # the SIDD stack, as eval hands it over


def iter_crop(dn):
    dn = np.asarray(dn)
    return dn[:, :128], dn[96:160, 4000:4200], dn[5::16, 3::16]


@pytest.mark.parametrize("ci", range(8))
def test_iter_denoise(golden, ci):
    """Row Q against the reference's own IterDenoise: cases 0-1 end at the beta1 < 0 guard (one output), 2-3 take the
    beta2 < 0 -> beta1**2 branch and CONTINUE, 4-7 the plain second round (two outputs, two regs rows)."""
    g = golden("iter")
    lr, clean, full, arch, sd, pipe = iter_case(g, ci)
    torch.set_num_threads(8)
    res = O.IterDenoise(lr, arch, sd, pipe, lr_full=full)
    regs = g[f"regs_{ci}"]
    assert len(res['regs']) == len(regs) and len(res['raw_dns']) == int(g[f"nout_{ci}"])
    assert len(regs) == (1 if str(g[f"weights_{ci}"]) == "random" else 2)
    if ci in (2, 3):
        assert regs[1][1] == regs[1][0] ** 2                     # the fixture really went through :438-440
    for r, gr in zip(res['regs'], regs):
        np.testing.assert_allclose(r[0], gr[0], rtol=1e-5)
        np.testing.assert_allclose(r[1], gr[1], rtol=0, atol=1e-5 * abs(gr[0]) + 1e-9)
    for it, dn in enumerate(res['raw_dns']):
        for got, tag in zip(iter_crop(dn), ("blk", "seam", "sub")):
            np.testing.assert_allclose(got, g[f"dn_{ci}_{it}_{tag}"], rtol=0, atol=2e-5)
        chk = g[f"dn_{ci}_{it}_chk"]
        np.testing.assert_allclose(np.asarray(dn, np.float64).sum(), chk[0], rtol=1e-6)


def test_bias_lut_2d(golden):
    """Row H': the oracle's BiasLUT against the reference's BiasLUT.get_lut on a small table built with the reference's
    get_bias_points -- table regenerated bit for bit by the oracle's restatement, lookups equal, VST_Denoiser through it."""
    g = golden("biaslut")
    x_lut, sg_lut, table = g["x_lut"], g["sg_lut"], g["table"]
    col = O.get_bias_points(x_lut.copy(), 1.0, float(sg_lut[5]), pho_min=20, close_form=True)
    np.testing.assert_allclose(col, table[:, 5], rtol=0, atol=1e-12)
    lut = O.BiasLUT(table, x_lut, sg_lut)
    for ci in range(int(g["ncases"])):
        K, s = (np.float64(v) for v in g[f"ksig_{ci}"])
        got = lut.get_lut(g[f"x_{ci}"].copy(), K=K, sigGs=s)
        np.testing.assert_allclose(got, g[f"bias_{ci}"], rtol=1e-12, atol=1e-12)
    noisy, _ = O.synth_noisy(96, 128, 4.37, 6.27, 55)
    assert np.array_equal(sha(noisy), g["sha_vd"])
    arch = ARCHS["gru8"]
    p = {'wp': 1023, 'bl': 64, 'ratio': 1, 'scale': 959.0, 'gain': np.float64(4.37), 'sigma': np.float64(6.27)}
    dn = O.VST_Denoiser(noisy, p, arch, O.procedural_state_dict(arch, 91), bias_corr='pre', biaslut=lut)
    np.testing.assert_allclose(dn, g["dn_vd"], rtol=0, atol=5e-6)


def test_ssim_against_reference_arithmetic(golden):
    """N1: the reference's own calculate_ssim (YOND_SIDD.py:679-721), run around a documented-semantics stub of
    cv2.getGaussianKernel / filter2D, against the oracle's restatement (the filter itself stays unpinned)."""
    g = golden("ssim")
    for ci in range(3):
        K, s = g[f"ksig_{ci}"]
        noisy, clean = O.synth_noisy(256, 512, K, s, 61 + ci)
        dn = np.clip(clean + 0.3 * (noisy - clean), 0, 1).astype(np.float32)
        vals = [O.ssim(a * 255, b * 255) for a, b in zip(np.split(dn, 2, axis=-1), np.split(clean, 2, axis=-1))]
        np.testing.assert_allclose(vals, g[f"ssim_{ci}"], rtol=1e-12)


def rot_case():
    K, s = 2.0, 20.0
    noisy, clean = O.synth_noisy(256, 8192, K, s, 31)
    full, _ = O.synth_noisy(512, 1024, K, s, 32)
    arch = ARCHS["gru8"]
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': False}
    p = dict(O.default_params(), rot_cfa=True, cfa=[[2, 1], [3, 2]])
    return np.array(np.split(noisy, 32, axis=-1)), full, arch, O.denoising_state_dict(arch, 81), pipe, p


def test_rot_bayer_and_rot_cfa_pipeline(golden):
    """N3: rot_bayer for the four CFA patterns, and IterDenoise with p['rot_cfa'] (YOND_SIDD.py:402-404, 462-464) against
    the reference's own run."""
    from yond_public_amd.utils.sidd_utils import rot_bayer, rot_k
    g = golden("rot")
    a = g["a"]
    for i in range(4):
        pat = g[f"pat_{i}"].tolist()
        assert np.array_equal(O.rot_bayer(a, pat), g[f"fwd_{i}"]) and np.array_equal(O.rot_bayer(a, pat, rev=True), g[f"rev_{i}"])
        assert np.array_equal(rot_bayer(a, pat), g[f"fwd_{i}"]) and np.array_equal(rot_bayer(a, pat, rev=True), g[f"rev_{i}"])
        assert (rot_k(pat) + rot_k(pat, rev=True)) % 4 == 0
    lr, full, arch, sd, pipe, p = rot_case()
    torch.set_num_threads(8)
    res = O.IterDenoise(lr, arch, sd, pipe, lr_full=full, p=p)
    assert len(res['raw_dns']) == int(g["nout"]) == 2
    for r, gr in zip(res['regs'], g["regs"]):
        np.testing.assert_allclose(r[0], gr[0], rtol=1e-5)
    for it, dn in enumerate(res['raw_dns']):
        for got, tag in zip(iter_crop(dn), ("blk", "seam", "sub")):
            np.testing.assert_allclose(got, g[f"dn_{it}_{tag}"], rtol=0, atol=2e-5)


def test_iter_denoise_without_estimate_branch(golden):
    """YOND_SIDD.py:358-381 (full_est False, est_type without 'pge'): no noise estimate, every block through
    Simple_Denoiser, regs = (0, 0) -- against the reference's own run."""
    g = golden("rot")
    noisy, _ = O.synth_noisy(256, 8192, 2.0, 20.0, 31)
    arch = ARCHS["unet8"]
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': False, 'full_est': False,
            'est_type': 'simple'}
    torch.set_num_threads(8)
    res = O.IterDenoise(np.array(np.split(noisy, 32, axis=-1)), arch, O.denoising_state_dict(arch, 82), pipe)
    assert res['regs'] == (0, 0) and len(res['raw_dns']) == 1
    for got, tag in zip(iter_crop(res['raw_dns'][0]), ("blk", "seam", "sub")):
        np.testing.assert_allclose(got, g[f"simple_{tag}"], rtol=0, atol=2e-5)


def iter_full_case():
    """The bare full frame of tests/golden/iter_full.npz (oracle/gen_golden.py iter_full_case): inputs are regenerated from seeds."""
    noisy, clean = O.synth_noisy(320, 2048, 2.0, 20.0, 41)
    arch = ARCHS["gru8"]
    sd = O.denoising_state_dict(arch, 91)
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': True}
    return noisy, clean, arch, sd, pipe


def iter_full_crop(dn):
    dn = np.asarray(dn)
    return dn[:96, :160].astype(np.float32), dn[200:264, 1000:1100].astype(np.float32), dn[3::8, 5::8].astype(np.float32)


def test_iter_denoise_bare_full_frame(golden):
    """What a bare full-frame 'iter' run equals: the reference's own IterDenoise (full_dn) on a 320 x 2048 frame, packed width
    1024 = 32 x 32, so its hard-coded SIDD_256 re-tiling of the collaborative estimate (:431) RUNS -- and so does the oracle's
    by default (no collab_sidd256 key in the pipe)."""
    g = golden("iter_full")
    noisy, clean, arch, sd, pipe = iter_full_case()
    assert np.array_equal(sha(noisy), g["sha"])
    torch.set_num_threads(8)
    res = O.IterDenoise(noisy, arch, sd, pipe)
    assert len(res['raw_dns']) == int(g["nout"]) == 2
    for r, gr in zip(res['regs'], g["regs"]):
        np.testing.assert_allclose(r[0], gr[0], rtol=1e-5)
        np.testing.assert_allclose(r[1], gr[1], rtol=0, atol=1e-5 * abs(gr[0]) + 1e-9)
    for it, dn in enumerate(res['raw_dns']):
        for got, tag in zip(iter_full_crop(dn), "abc"):
            np.testing.assert_allclose(got, g[f"dn_{it}_{tag}"], rtol=0, atol=2e-5)
    # without the re-tiling the second estimate is a different number: the fixture does pin the branch
    res2 = O.IterDenoise(noisy, arch, sd, dict(pipe, collab_sidd256=False))
    assert abs(res2['regs'][1][0] - g["regs"][1][0]) > 1e-7 * abs(g["regs"][1][0])


def test_nle_full_frame_3000x4000(golden):
    """SURVEY section 8c plan item 3: the oracle's estimator on the cfg-2 frame against the reference's eight numbers."""
    g = golden("nle_full")
    H, W, K, s, idx = g["meta"]
    noisy, clean = O.synth_noisy(int(H), int(W), K, s, int(idx))
    assert np.array_equal(sha(noisy), g["sha"])
    reg, info = O.SimpleNLF(noisy, k=29, setting={'mode': 'self'}, full=True)
    th, pct, b1, b2 = g["self"]
    assert info['percent'] == pct
    np.testing.assert_allclose(info['th'], th, rtol=1e-6)
    np.testing.assert_allclose(reg[0], b1, rtol=1e-6)
    np.testing.assert_allclose(reg[1], b2, rtol=0, atol=1e-6 * abs(b1) + 1e-10)
    dn = np.clip(clean + 0.002 * np.sin(np.arange(int(W))[None, :] / 37.0), 0, 1).astype(np.float32)
    regc, infoc = O.SimpleNLF(noisy, dn, k=29, setting={'mode': 'collab'}, full=True)
    thc, pctc, c1, c2 = g["collab"]
    assert infoc['percent'] == pctc
    np.testing.assert_allclose(infoc['th'], thc, rtol=1e-6)
    np.testing.assert_allclose(regc[0], c1, rtol=1e-6)
    np.testing.assert_allclose(regc[1], c2, rtol=0, atol=1e-6 * abs(c1) + 1e-10)


# ---- BASELINE.json's configurations at their real sizes (tests/golden/full_cfg{2,4,5}.npz, oracle/gen_golden.py gen_full_*) ----
FULL_PIPE = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'max_iter': 1, 'full_dn': True}


def full_crops(dn):
    dn = np.asarray(dn)
    H, W = dn.shape
    return (dn[:64, :64].astype(np.float32), dn[H // 2 - 32:H // 2 + 32, W // 2 - 32:W // 2 + 32].astype(np.float32),
            dn[H - 64:, W - 64:].astype(np.float32), dn[7::64, 11::64].astype(np.float32))


def full_case(which, i=0):
    """Inputs of the full-size fixtures, regenerated from seeds (the generating script's full_cfg*_case functions)."""
    if which == "cfg2":
        noisy, clean = O.synth_noisy(3000, 4000, 4.0, 6.0, 0)
        arch, seed = ARCHS["gru32"], 7
    elif which == "cfg2w":
        noisy, clean = O.synth_noisy(3000, 4096, 2.0, 20.0, 3)
        arch, seed = ARCHS["gru32"], 8
    elif which == "cfg4":
        noisy, clean = O.synth_noisy(3000, 4000, 4.0, 6.0, 70 + i)
        arch, seed = ARCHS["unet32"], 9
    else:
        rng = np.random.default_rng(1997 + 55)
        clean = (O.synth_clean(4000, 6000) * 0.2).astype(np.float32)
        noisy = ((rng.poisson(clean * 959.0 / 2.0) * 2.0 + rng.normal(0.0, 25.0, clean.shape)) / 959.0).astype(np.float32)
        arch, seed = ARCHS["gru32"], 10
    return noisy, clean, arch, O.denoising_state_dict(arch, seed)


def test_full_size_fixture_inputs_regenerate(golden):
    """The full-size fixtures hold outputs only; their inputs come back bit for bit from the seeds, and the reference's round-1
    estimate on the 3000 x 4000 frame inside IterDenoise is the number its stand-alone SimpleNLF gave (nle_full.npz)."""
    g2, g4, g5 = golden("full_cfg2"), golden("full_cfg4"), golden("full_cfg5")
    assert np.array_equal(sha(full_case("cfg2")[0]), g2["a_sha"])
    assert np.array_equal(sha(full_case("cfg2w")[0]), g2["b_sha"])
    for i in range(2):
        assert np.array_equal(sha(full_case("cfg4", i)[0]), g4[f"sha_{i}"])
    n5 = full_case("cfg5")[0]
    assert np.array_equal(sha(n5), g5["sha"]) and float(n5.min()) < 0
    np.testing.assert_allclose(g2["a_reg0"], golden("nle_full")["self"][2:], rtol=1e-12)
    assert g2["a_psnr"][1] > g2["a_psnr"][0] + 7 and int(g2["b_nout"]) == 2


@pytest.mark.skipif(not os.environ.get("YOND_SLOW"), reason="minutes of CPU: the oracle's full-size forward (YOND_SLOW=1)")
def test_oracle_full_cfg2_round1(golden):
    g = golden("full_cfg2")
    noisy, clean, arch, sd = full_case("cfg2")
    torch.set_num_threads(8)
    res = O.IterDenoise(noisy, arch, sd, dict(FULL_PIPE, iter='once'))
    np.testing.assert_allclose(res['regs'][0], g["a_reg0"], rtol=1e-5)
    for got, tag in zip(full_crops(res['raw_dns'][0]), ("a", "b", "c", "sub")):
        np.testing.assert_allclose(got, g[f"a_dn0_{tag}"], rtol=0, atol=2e-5)
