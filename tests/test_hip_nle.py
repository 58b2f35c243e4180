"""GPU: noise-level estimation kernels (K5 box statistics, K6 percentiles, K7 accumulation) and the
assembled SimpleNLF against the oracle and the reference's golden vectors.
Tolerances: th, beta1 rel <= 1e-5; beta2 abs <= 1e-5*beta1 + 1e-9."""
import numpy as np
import pytest
import torch

from hip_common import report, sha

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def planes(a):
    """oracle HWC (h, w, 4) -> planar (4, h, w)"""
    return np.ascontiguousarray(np.transpose(a, (2, 0, 1)))


@pytest.mark.parametrize("H,W", [(256, 256), (200, 328), (62, 70)])
def test_box_stats_self(H, W):
    import yond_oracle as O
    from yond_public_amd import _lib as L
    lib = L.load()
    noisy, _ = O.synth_noisy(H, W, 4.0, 6.0, 11)
    rggb = O.bayer2rggb(noisy)
    mean = O.box_blur(rggb, 29)
    var = O.stdfilt(rggb, 29) ** 2
    b2 = O.box_blur(rggb, 19)
    lap = O.stdfilt(b2, 29)
    h, w = H // 2, W // 2
    t = torch.from_numpy(noisy).to(DEV)
    o = [torch.empty((4, h, w), device=DEV) for _ in range(4)]
    L.check(lib.yond_box_stats_self1_f32(L.ptr(t), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.stream()), "self1")
    L.check(lib.yond_box_stats_self2_f32(L.ptr(o[2]), h, w, 29, 0, L.ptr(o[3]), L.stream()), "self2")
    got = [x.cpu().numpy() for x in o]
    # window sums of float32 data are exact in float64, so the maps agree to the bit except where the
    # oracle's cumulative-sum order flips a float32 rounding (<= 1 ulp)
    assert report("mean", got[0], planes(mean)) <= 6e-8
    assert report("blur19", got[2], planes(b2)) <= 6e-8
    assert report("var", got[1], planes(var)) <= 2e-8
    assert report("lap", got[3], planes(lap)) <= 2e-6
    frac = np.mean(got[0] == planes(mean))
    print("bit-identical mean fraction", frac)
    assert frac > 0.99


def test_box_stats_collab_and_sidd_tiling():
    import yond_oracle as O
    from yond_public_amd import _lib as L
    lib = L.load()
    H, W = 256, 2048
    noisy, clean = O.synth_noisy(H, W, 4.0, 6.0, 12)
    dn = np.clip(clean + 0.002 * np.sin(np.arange(W)[None, :] / 37.0), 0, 1).astype(np.float32)
    for tile_w in (0, 64):
        lr, hr = O.bayer2rggb(noisy), O.bayer2rggb(dn)
        if tile_w:
            nt = (W // 2) // tile_w
            lr = np.concatenate(np.split(lr, nt, axis=-2), axis=-1)
            hr = np.concatenate(np.split(hr, nt, axis=-2), axis=-1)
        lr_k, hr_k = O.stdfilt(lr, 29), O.stdfilt(hr, 29)
        var, mean, lap = lr_k ** 2 - hr_k ** 2, O.box_blur(hr, 29), hr_k
        if tile_w:  # back to (h, w, 4)
            un = lambda a: np.concatenate(np.split(a, nt, axis=-1), axis=-2)
            var, mean, lap = un(var), un(mean), un(lap)
        h, w = H // 2, W // 2
        o = [torch.empty((4, h, w), device=DEV) for _ in range(3)]
        nd, dd = torch.from_numpy(noisy).to(DEV), torch.from_numpy(dn).to(DEV)
        L.check(lib.yond_box_stats_collab_f32(L.ptr(nd), L.ptr(dd), H, W, 29,
                                              tile_w, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.stream()), "collab")
        got = [x.cpu().numpy() for x in o]
        assert report(f"collab mean tile_w={tile_w}", got[0], planes(mean)) <= 6e-8
        assert report(f"collab var tile_w={tile_w}", got[1], planes(var)) <= 6e-8
        assert report(f"collab lap tile_w={tile_w}", got[2], planes(lap)) <= 2e-6


@pytest.mark.parametrize("n", [1, 5, 1000, 65537, 3_000_001])
def test_percentiles_exact(n):
    from yond_public_amd import pipeline as P
    rng = np.random.default_rng(n)
    a = (rng.random(n).astype(np.float32) ** 3) * 0.1
    if n > 100:
        a[:: 7] = a[3]                      # heavy duplicates
    q = np.linspace(5, 100, 20)
    got = P._percentiles(torch.from_numpy(a).to(DEV), q).cpu().numpy()
    ref = np.percentile(a, q, method='linear')
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    got25 = P._percentiles(torch.from_numpy(a).to(DEV), [25.0]).cpu().numpy()
    assert np.array_equal(got25, np.percentile(a, [25.0], method='linear'))


def test_select_handles_negative_and_zero_values():
    from yond_public_amd import pipeline as P
    rng = np.random.default_rng(2)
    a = rng.standard_normal(10007).astype(np.float32)
    a[:50] = 0.0
    a[50:60] = -0.0
    q = [0.0, 12.5, 50.0, 99.9, 100.0]
    got = P._percentiles(torch.from_numpy(a).to(DEV), q).cpu().numpy()
    assert np.array_equal(got, np.percentile(a, q, method='linear'))


@pytest.mark.parametrize("off", [0, 1, 2, 3])
def test_select_unaligned_large_and_constant(off):
    """Edge cases of the three-level select: a data pointer that is not 16-byte aligned, values outside the
    [0, 2) LDS window (negative, large), and constant data (every element in one bin)."""
    from yond_public_amd import pipeline as P
    rng = np.random.default_rng(30 + off)
    n = 70001
    a = (rng.standard_normal(n + off) * np.float32(3.0)).astype(np.float32)
    a[100:200] = 1e6
    a[200:260] = -1e-30
    t = torch.from_numpy(a).to(DEV)[off:]
    q = [0.0, 1.0, 33.3, 50.0, 75.0, 99.99, 100.0]
    got = P._percentiles(t, q).cpu().numpy()
    assert np.array_equal(got, np.percentile(a[off:], q, method='linear'))
    c = torch.full((50003 + off,), 0.0123, device=DEV)[off:]
    got = P._percentiles(c, q).cpu().numpy()
    assert np.array_equal(got, np.full(len(q), np.float64(np.float32(0.0123))))


@pytest.mark.parametrize("shape", [(1_000_003,), (750, 1336), (333, 1001)])
def test_occupancy_score3_moments_match_numpy(shape):
    """K7a / K7s / K7b against NumPy: 1-D data (one row), a width that is a multiple of 4 (vector path) and an
    odd width (scalar path)."""
    from yond_public_amd import pipeline as P
    rng = np.random.default_rng(4)
    n = int(np.prod(shape))
    lap = (rng.random(n).astype(np.float32) ** 2) * 0.05
    mean = np.clip(rng.random(n).astype(np.float32) * 1.1 - 0.05, -0.02, 1.05).astype(np.float32)
    var = (0.004 * mean + 1e-4 * rng.standard_normal(n)).astype(np.float32)
    q = np.linspace(5, 100, 20)
    ths = np.percentile(lap, q, method='linear')
    lap[:3] = ths[4].astype(np.float32)                   # values exactly on / next to a threshold
    width = shape[-1] if len(shape) > 1 else None
    d = lambda a: torch.from_numpy(a).to(DEV)
    ths_dev = d(ths)
    occ = P._occupancy(d(lap), d(mean), ths_dev, width)
    seen = np.logical_or.accumulate(P._occ_unpack(occ.cpu().numpy()), axis=0).sum(axis=1)
    ref_np = np.zeros(20, np.int64)
    for i in range(20):
        b = (mean[lap <= ths[i]].clip(0, 1) * 1000).astype(int)
        ref_np[i] = np.sum(np.bincount(b, minlength=1001) > 0)
    assert np.array_equal(seen, ref_np)
    sel, npk = P._score3_device(occ, ths_dev, q)
    sel, npk = sel.cpu().numpy(), npk.cpu().numpy()
    assert np.array_equal(npk, ref_np)
    score = ths / (q * ref_np.astype(np.float64))
    i = int(np.argmin(score[1:]) + 1)
    assert int(sel[0]) == i and sel[1] == ths[i] and sel[2] == q[i] and sel[3] == score[i]
    for i in (0, 4, 5, 19):
        mom = P._moments(d(lap), d(mean), d(var), ths_dev[i:i + 1]).cpu().numpy()
        pick = lap < ths[i]
        for j, s_ in enumerate((pick, pick & (mean > np.float32(1e-4)) & (mean < np.float32(0.8)))):
            m, v = mean[s_].astype(np.float64), var[s_].astype(np.float64)
            ref = np.array([m.size, m.sum(), v.sum(), (m * m).sum(), (m * v).sum()])
            np.testing.assert_allclose(mom[j], ref, rtol=1e-11)
    inf = torch.full((1,), float('inf'), dtype=torch.float64, device=DEV)
    assert P._moments(d(lap), d(mean), d(var), inf).cpu().numpy()[0, 0] == n


@pytest.mark.parametrize("tag", ["s256", "s512", "hi", "lo"])
def test_simple_nlf_matches_reference(golden, tag):
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    g = golden("nle")
    H, W, K, s, idx = g[f"{tag}_meta"]
    noisy, clean = O.synth_noisy(int(H), int(W), K, s, int(idx))
    assert np.array_equal(sha(noisy), g[f"{tag}_sha"])
    reg, info = P.SimpleNLF(noisy, k=29, setting={'mode': 'self'}, full=True, device=DEV)
    th, pct, b1, b2 = g[f"{tag}_self"]
    print(f"[parity] NLE self {tag}: th {info['th']:.6e} vs {th:.6e}; b1 {reg[0]:.6e} vs {b1:.6e}; b2 {reg[1]:.6e} vs {b2:.6e}")
    assert info['percent'] == pct
    np.testing.assert_allclose(info['th'], th, rtol=1e-5)
    np.testing.assert_allclose(reg[0], b1, rtol=1e-5)
    np.testing.assert_allclose(reg[1], b2, rtol=0, atol=1e-5 * abs(b1) + 1e-9)
    dn = np.clip(clean + 0.002 * np.sin(np.arange(int(W))[None, :] / 37.0), 0, 1).astype(np.float32)
    regc, infoc = P.SimpleNLF(noisy, dn, k=29, setting={'mode': 'collab'}, full=True, device=DEV)
    thc, pctc, c1, c2 = g[f"{tag}_collab"]
    assert infoc['percent'] == pctc
    np.testing.assert_allclose(infoc['th'], thc, rtol=1e-5)
    np.testing.assert_allclose(regc[0], c1, rtol=1e-5)
    np.testing.assert_allclose(regc[1], c2, rtol=0, atol=1e-5 * abs(c1) + 1e-9)


def test_simple_nlf_sidd256(golden):
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    g = golden("nle")
    noisy, clean = O.synth_noisy(256, 8192, 4.0, 6.0, 7)
    assert np.array_equal(sha(noisy), g["strip_sha"])
    dn = np.clip(clean + 0.002 * np.sin(np.arange(8192)[None, :] / 37.0), 0, 1).astype(np.float32)
    r1 = P.SimpleNLF(noisy, k=29, setting={'mode': 'self', 'SIDD_256': True}, device=DEV)
    r2 = P.SimpleNLF(noisy, dn, k=29, setting={'mode': 'collab', 'SIDD_256': True}, device=DEV)
    r3 = P.SimpleNLF(noisy, k=29, setting={'mode': 'self'}, device=DEV)
    for r, gr in zip((r1, r2, r3), g["strip_regs"]):
        np.testing.assert_allclose(r[0], gr[0], rtol=1e-5)
        np.testing.assert_allclose(r[1], gr[1], rtol=0, atol=1e-5 * abs(gr[0]) + 1e-9)


def test_simple_nlf_full_frame_matches_reference(golden):
    """SURVEY section 8c plan item 3: the estimator on the BASELINE cfg-2 frame itself (3000 x 4000) against the eight numbers
    the reference's own SimpleNLF / get_threshold produce for it (tests/golden/nle_full.npz): self and collaborative."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    g = golden("nle_full")
    H, W, K, s, idx = g["meta"]
    noisy, clean = O.synth_noisy(int(H), int(W), K, s, int(idx))
    assert np.array_equal(sha(noisy), g["sha"])
    reg, info = P.SimpleNLF(noisy, k=29, setting={'mode': 'self'}, full=True, device=DEV)
    th, pct, b1, b2 = g["self"]
    print(f"[parity] NLE self 3000x4000: th {info['th']:.8e} vs {th:.8e}; b1 {reg[0]:.8e} vs {b1:.8e}; b2 {reg[1]:.8e} vs {b2:.8e}")
    assert info['percent'] == pct
    np.testing.assert_allclose(info['th'], th, rtol=1e-5)
    np.testing.assert_allclose(reg[0], b1, rtol=1e-5)
    np.testing.assert_allclose(reg[1], b2, rtol=0, atol=1e-5 * abs(b1) + 1e-9)
    dn = np.clip(clean + 0.002 * np.sin(np.arange(int(W))[None, :] / 37.0), 0, 1).astype(np.float32)
    regc, infoc = P.SimpleNLF(noisy, dn, k=29, setting={'mode': 'collab'}, full=True, device=DEV)
    thc, pctc, c1, c2 = g["collab"]
    print(f"[parity] NLE collab 3000x4000: th {infoc['th']:.8e} vs {thc:.8e}; b1 {regc[0]:.8e} vs {c1:.8e}; b2 {regc[1]:.8e} vs {c2:.8e}")
    assert infoc['percent'] == pctc
    np.testing.assert_allclose(infoc['th'], thc, rtol=1e-5)
    np.testing.assert_allclose(regc[0], c1, rtol=1e-5)
    np.testing.assert_allclose(regc[1], c2, rtol=0, atol=1e-5 * abs(c1) + 1e-9)


def test_nlf_full_frame_size_properties():
    """BASELINE cfg-2 size (3000 x 4000): size-independent checks -- percentiles are sorted and bracket the
    data, bucket counts add up to the pixel count, the estimate recovers the synthetic (K, sigma)."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    noisy, _ = O.synth_noisy(3000, 4000, 4.0, 6.0, 0)
    reg, info = P.SimpleNLF(noisy, k=29, setting={'mode': 'self'}, full=True, device=DEV)
    ths = info['ths']
    assert np.all(np.diff(ths) >= 0)
    K, sig = reg[0] * 959, np.sqrt(max(reg[1], 0)) * 959
    print(f"[cfg2] estimated K={K:.4f} sigma={sig:.4f} (synthetic 4.0 / 6.0), percent={info['percent']}")
    assert abs(K - 4.0) < 0.2 and abs(sig - 6.0) < 3.0


@pytest.mark.parametrize("shape", [(5, 1), (62, 70), (1000, 13), (4, 333, 500), (4, 1500, 2000)])
def test_two_sweep_threshold_selection(shape):
    """K6'/K7' (nle_fast.hip): percentiles bit-equal to np.percentile, npeaks and the selected index equal to the
    reference formula (YOND_SIDD.py:22-49) -- odd widths, unaligned sizes, heavy duplicates, values outside [0, 2)."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    rng = np.random.default_rng(sum(shape))
    n = int(np.prod(shape))
    lap = ((rng.random(n).astype(np.float32) ** 3) * 0.05).reshape(shape)
    # smooth component so that runs of equal level-1 bins occur, plus duplicates and a few outliers
    lap = (lap + np.float32(0.01) * np.linspace(0, 1, shape[-1], dtype=np.float32)).astype(np.float32)
    flat = lap.reshape(-1)
    if n > 100:
        flat[::7] = flat[3]
        flat[5] = 3.5
        flat[6] = 1e-30
    mean = (rng.random(n).astype(np.float32) * 1.2 - 0.1).reshape(shape)
    quants = np.linspace(5, 100, 20)
    ths, npeaks, sel, _ = P._threshold_state(torch.from_numpy(lap).to(DEV), torch.from_numpy(mean).to(DEV), quants)
    ref = np.percentile(flat, quants, method='linear')
    assert np.array_equal(ths, ref), np.abs(ths - ref).max()
    th, pct, info = O.get_threshold_score3(lap.reshape(-1), mean.reshape(-1), step=5, full=True)
    np.testing.assert_array_equal(npeaks, info['npeaks'].astype(np.int64))
    assert int(sel[0]) == info['index'] and sel[1] == th and sel[2] == pct


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_two_sweep_selection_wide_dynamic_range(seed):
    """The level-1 histogram is stored transposed inside blocks of 4096 bins and the resolve reads only the blocks the
    writers marked: data that spreads over MANY blocks (negative values, denormals, 1e-30 ... 1e30, +-0, exact
    duplicates across block borders) must still give np.percentile bit for bit, and the reference's npeaks / index."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    rng = np.random.default_rng(100 + seed)
    shape = (257, 1003) if seed else (64, 4096)
    n = int(np.prod(shape))
    expo = rng.uniform(-30, 30, n) if seed < 2 else rng.uniform(-3, 1, n)
    lap = (np.float32(10.0) ** expo.astype(np.float32)).astype(np.float32)
    lap[rng.random(n) < 0.2] *= np.float32(-1.0)                     # keys below 0x80000000: the lower eight blocks
    lap[rng.random(n) < 0.05] = 0.0
    lap[rng.random(n) < 0.01] = -0.0
    lap[::11] = np.float32(2.0) ** np.float32(-14)                   # first bin of a block (key 0xB880 0000 >> 16 = 0xB880)
    lap[1::13] = np.nextafter(np.float32(2.0) ** np.float32(-14), np.float32(0))   # last bin of the block below
    lap[5] = np.float32(1e-42)                                        # a denormal
    lap = lap.reshape(shape)
    mean = (rng.random(n).astype(np.float32) * 1.2 - 0.1).reshape(shape)
    quants = np.linspace(5, 100, 20)
    ths, npeaks, sel, _ = P._threshold_state(torch.from_numpy(lap).to(DEV), torch.from_numpy(mean).to(DEV), quants)
    ref = np.percentile(lap.reshape(-1), quants, method='linear')
    assert np.array_equal(ths, ref), (ths, ref)
    th, pct, info = O.get_threshold_score3(lap.reshape(-1), mean.reshape(-1), step=5, full=True)
    np.testing.assert_array_equal(npeaks, info['npeaks'].astype(np.int64))
    assert int(sel[0]) == info['index'] and sel[1] == th and sel[2] == pct


def test_two_sweep_selection_constant_and_full_frame_property():
    """Degenerate data (one level-1 bin holds everything: every element is a candidate) and, at the full cfg-2 size, a
    size-independent property: the selected threshold splits the data at its quantile."""
    from yond_public_amd import pipeline as P
    quants = np.linspace(5, 100, 20)
    c = torch.full((300, 1001), 0.0123, device=DEV)
    m = torch.rand((300, 1001), device=DEV)
    ths, npeaks, sel, _ = P._threshold_state(c, m, quants)
    assert np.array_equal(ths, np.full(20, np.float64(np.float32(0.0123))))
    assert npeaks.min() == npeaks.max() == len(torch.unique((m.clamp(0, 1) * 1000).int()))
    g = torch.Generator(device=DEV).manual_seed(1)
    lap = torch.rand((4, 1500, 2000), device=DEV, generator=g) ** 2 * 0.02
    mean = torch.rand((4, 1500, 2000), device=DEV, generator=g)
    ths, npeaks, sel, _ = P._threshold_state(lap, mean, quants)
    n = lap.numel()
    for q, t in zip(quants, ths):
        below = int((lap.double() < t).sum().item())
        le = int((lap.double() <= t).sum().item())
        r = q / 100.0 * (n - 1)
        assert below <= r + 1 and le >= r - 1, (q, t, below, le, r)


@pytest.mark.parametrize("H,W,tile_w", [(256, 256, 0), (200, 328, 0), (62, 70, 0), (700, 1000, 0), (256, 2048, 32)])
def test_fused_box_statistics_self(H, W, tile_w):
    """K5' (nle_fused.hip): one pass -> mean, var, lap (the B19 map never leaves the chip) + the level-1 statistics of
    the threshold selection + the frame maximum.  Maps against the oracle (<= 1 ulp) and BIT-identical to the
    stand-alone kernels; the selection state against the stand-alone sweep."""
    if not __import__("yond_public_amd._lib", fromlist=["has"]).has("yond_box_stats_self_fused_f32"):
        pytest.skip("the one-pass kernels exist only in experiment builds (python -m yond_public_amd.build --experiments)")
    import yond_oracle as O
    from yond_public_amd import _lib as L
    from yond_public_amd import pipeline as P
    lib = L.load()
    noisy, _ = O.synth_noisy(H, W, 4.0, 6.0, 11)
    h, w = H // 2, W // 2
    t = torch.from_numpy(noisy).to(DEV)
    o = [torch.empty((4, h, w), device=DEV) for _ in range(7)]
    q = np.ascontiguousarray(P.QUANTS)
    import ctypes as C
    qp = C.c_void_p(q.ctypes.data)
    ws = P._nle_workspace(4 * h * w, t.device)
    L.check(lib.yond_box_stats_self_fused_f32(L.ptr(t), H, W, 29, 19, tile_w, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), qp, len(q),
                                              L.ptr(ws), L.stream()), "fused")
    L.check(lib.yond_box_stats_self1_f32(L.ptr(t), H, W, 29, 19, tile_w, L.ptr(o[3]), L.ptr(o[4]), L.ptr(o[6]), L.stream()), "self1")
    L.check(lib.yond_box_stats_self2_f32(L.ptr(o[6]), h, w, 29, tile_w, L.ptr(o[5]), L.stream()), "self2")
    for a, b, name in zip(o[:3], o[3:6], ("mean", "var", "lap")):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    if not tile_w:
        rggb = O.bayer2rggb(noisy)
        assert report("fused mean", o[0].cpu().numpy(), planes(O.box_blur(rggb, 29))) <= 6e-8
        assert report("fused var", o[1].cpu().numpy(), planes(O.stdfilt(rggb, 29) ** 2)) <= 2e-8
        assert report("fused lap", o[2].cpu().numpy(), planes(O.stdfilt(O.box_blur(rggb, 19), 29))) <= 2e-6
    # selection state: finish on the fused workspace and on a stand-alone one
    ths_f, np_f, sel_f, _ = P._threshold_state(o[2], o[0], q, ws=ws)
    ths_s, np_s, sel_s, _ = P._threshold_state(o[5], o[3], q)
    assert np.array_equal(ths_f, ths_s) and np.array_equal(np_f, np_s) and np.array_equal(sel_f, sel_s)
    assert np.array_equal(ths_f, np.percentile(o[2].cpu().numpy().reshape(-1), q, method='linear'))
    off_max = P._nle_layout()[4]
    key = int(ws[off_max:off_max + 4].cpu().numpy().view(np.uint32)[0])
    assert P._key2float(key) == noisy.max()


def test_fused_box_statistics_collab():
    if not __import__("yond_public_amd._lib", fromlist=["has"]).has("yond_box_stats_self_fused_f32"):
        pytest.skip("the one-pass kernels exist only in experiment builds (python -m yond_public_amd.build --experiments)")
    import ctypes as C
    import yond_oracle as O
    from yond_public_amd import _lib as L
    from yond_public_amd import pipeline as P
    lib = L.load()
    H, W = 256, 2048
    noisy, clean = O.synth_noisy(H, W, 4.0, 6.0, 12)
    dn = np.clip(clean + 0.002 * np.sin(np.arange(W)[None, :] / 37.0), 0, 1).astype(np.float32)
    q = np.ascontiguousarray(P.QUANTS)
    qp = C.c_void_p(q.ctypes.data)
    h, w = H // 2, W // 2
    nd, dd = torch.from_numpy(noisy).to(DEV), torch.from_numpy(dn).to(DEV)
    for tile_w in (0, 64):
        o = [torch.empty((4, h, w), device=DEV) for _ in range(6)]
        ws = P._nle_workspace(4 * h * w, nd.device)
        L.check(lib.yond_box_stats_collab_fused_f32(L.ptr(nd), L.ptr(dd), H, W, 29, tile_w, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]),
                                                    qp, len(q), L.ptr(ws), L.stream()), "collab fused")
        L.check(lib.yond_box_stats_collab_f32(L.ptr(nd), L.ptr(dd), H, W, 29, tile_w, L.ptr(o[3]), L.ptr(o[4]), L.ptr(o[5]),
                                              L.stream()), "collab")
        # the sliding window sums are exact; the prefix scans of the stand-alone kernel carry ~1e-15 relative error, which
        # now and then flips a float32 rounding: agreement to one ulp, and both within the oracle's tolerance
        for a, b, name, tol in zip(o[:3], o[3:], ("mean", "var", "lap"), (6e-8, 1e-9, 1e-7)):
            d = float((a - b).abs().max())
            frac = float((a == b).float().mean())
            print(f"[parity] collab fused vs stand-alone {name} tile_w={tile_w}: max diff {d:.3e}, identical {frac:.6f}")
            assert d <= tol and frac > 0.999, (name, tile_w, d, frac)
        lr, hr = O.bayer2rggb(noisy), O.bayer2rggb(dn)
        if tile_w:
            nt = w // tile_w
            lr = np.concatenate(np.split(lr, nt, axis=-2), axis=-1)
            hr = np.concatenate(np.split(hr, nt, axis=-2), axis=-1)
        lr_k, hr_k = O.stdfilt(lr, 29), O.stdfilt(hr, 29)
        var, mean, lap = lr_k ** 2 - hr_k ** 2, O.box_blur(hr, 29), hr_k
        if tile_w:
            un = lambda a: np.concatenate(np.split(a, nt, axis=-1), axis=-2)
            var, mean, lap = un(var), un(mean), un(lap)
        assert report(f"collab fused mean tile_w={tile_w}", o[0].cpu().numpy(), planes(mean)) <= 6e-8
        assert report(f"collab fused var tile_w={tile_w}", o[1].cpu().numpy(), planes(var)) <= 6e-8
        assert report(f"collab fused lap tile_w={tile_w}", o[2].cpu().numpy(), planes(lap)) <= 2e-6
        # selection state of the fused pass against the stand-alone sweep over the SAME maps
        ths_f, np_f, sel_f, _ = P._threshold_state(o[2], o[0], q, ws=ws)
        ths_s, np_s, sel_s, _ = P._threshold_state(o[2], o[0], q)
        assert np.array_equal(ths_f, ths_s) and np.array_equal(np_f, np_s) and np.array_equal(sel_f, sel_s)


@pytest.mark.parametrize("H,W,tile_w,k", [(256, 256, 0, 29), (200, 328, 0, 29), (62, 70, 0, 13), (700, 1000, 0, 29), (256, 2048, 32, 29),
                                          (1024, 1024, 0, 21)])
def test_two_pass_box_statistics(H, W, tile_w, k):
    """K5+ (nle.hip, STATS): the streaming kernels with the level-1 statistics, the per-bin minimum of lap, the resolve of
    the ranks and the frame maximum folded in -- maps bit-identical to the plain kernels (same code), selection state
    identical to the stand-alone sweep, for self and collab, any window size."""
    import ctypes as C
    import yond_oracle as O
    from yond_public_amd import _lib as L
    from yond_public_amd import pipeline as P
    lib = L.load()
    noisy, clean = O.synth_noisy(H, W, 4.0, 6.0, 21)
    noisy[::7, ::5] = 0.0                                   # exact zeros: flat windows give lap == +0.0 (the side counter)
    h, w = H // 2, W // 2
    k2 = k // 3 * 2 + 1
    t, c = torch.from_numpy(noisy).to(DEV), torch.from_numpy(clean).to(DEV)
    q = np.ascontiguousarray(P.QUANTS)
    qp = C.c_void_p(q.ctypes.data)
    off_max = P._nle_layout()[4]
    st = L.stream()
    # self
    o = [torch.empty((4, h, w), device=DEV) for _ in range(8)]
    ws = P._nle_workspace(4 * h * w, t.device)
    L.check(lib.yond_box_stats_self_stats_f32(L.ptr(t), H, W, k, k2, tile_w, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[6]), L.ptr(o[2]),
                                              qp, len(q), L.ptr(ws), st), "self stats")
    L.check(lib.yond_box_stats_self1_f32(L.ptr(t), H, W, k, k2, tile_w, L.ptr(o[3]), L.ptr(o[4]), L.ptr(o[7]), st), "self1")
    L.check(lib.yond_box_stats_self2_f32(L.ptr(o[7]), h, w, k, tile_w, L.ptr(o[5]), st), "self2")
    for a, b, name in zip(o[:3], o[3:6], ("mean", "var", "lap")):
        assert torch.equal(a, b), name
    ths_f, np_f, sel_f, _ = P._threshold_state(o[2], o[0], q, ws=ws)
    ths_s, np_s, sel_s, _ = P._threshold_state(o[5], o[3], q)
    assert np.array_equal(ths_f, ths_s) and np.array_equal(np_f, np_s) and np.array_equal(sel_f, sel_s)
    assert np.array_equal(ths_f, np.percentile(o[2].cpu().numpy().reshape(-1), q, method='linear'))
    key = int(ws[off_max:off_max + 4].cpu().numpy().view(np.uint32)[0])
    assert P._key2float(key) == noisy.max()
    # collab
    L.check(lib.yond_box_stats_collab_stats_f32(L.ptr(t), L.ptr(c), H, W, k, tile_w, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), qp, len(q),
                                                L.ptr(ws), st), "collab stats")
    L.check(lib.yond_box_stats_collab_f32(L.ptr(t), L.ptr(c), H, W, k, tile_w, L.ptr(o[3]), L.ptr(o[4]), L.ptr(o[5]), st), "collab")
    for a, b, name in zip(o[:3], o[3:6], ("mean", "var", "lap")):
        assert torch.equal(a, b), name
    ths_f, np_f, sel_f, _ = P._threshold_state(o[2], o[0], q, ws=ws)
    ths_s, np_s, sel_s, _ = P._threshold_state(o[5], o[3], q)
    assert np.array_equal(ths_f, ths_s) and np.array_equal(np_f, np_s) and np.array_equal(sel_f, sel_s)


def test_simple_nlf_fused_equals_unfused_full_frame():
    """cfg-2 size: the three producers of the maps (two-pass with the statistics folded in -- the default --, the
    one-pass kernel, the stand-alone kernels + separate sweep) give the same estimate (same maps bit for bit; the float64
    moment sums are accumulated with atomics, so the fit agrees to the summation order)."""
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    noisy, _ = O.synth_noisy(3000, 4000, 4.0, 6.0, 3)
    t = torch.from_numpy(noisy).to(DEV)
    ra, ia = P.SimpleNLF(t, k=29, setting={'mode': 'self'}, full=True)
    rb, ib = P.SimpleNLF(t, k=29, setting={'mode': 'self'}, full=True, box='plain')
    from yond_public_amd import _lib as _L
    for box in (('one-pass', 'plain') if _L.has("yond_box_stats_self_fused_f32") else ('plain',)):
        rb, ib = P.SimpleNLF(t, k=29, setting={'mode': 'self'}, full=True, box=box)
        assert ia['th'] == ib['th'] and ia['percent'] == ib['percent'] and ia['nsel'] == ib['nsel'], box
        np.testing.assert_array_equal(ia['npeaks'], ib['npeaks'])
        np.testing.assert_allclose(ra, rb, rtol=1e-9)
        if box != 'plain':
            assert ib['frame_max'] == noisy.max()
    assert ia['frame_max'] == noisy.max()


def test_threshold_can_be_repeated_on_one_workspace():
    """yond_nle_threshold_f32 twice on the same sweep-1 state: sweep 2's fill counters and its arrival ticket are reset by the
    call's last workgroup, so the second call selects the same thresholds (an advisor's finding: it used to append its
    candidates behind the first call's and never fire the finishing workgroup)."""
    import ctypes as C
    from yond_public_amd import _lib as L
    from yond_public_amd import pipeline as P
    lib = L.load()
    rng = np.random.default_rng(11)
    h, w = 96, 160
    lap = torch.from_numpy((rng.random((4, h, w), dtype=np.float32) * 0.05)).to(DEV)
    mean = torch.from_numpy(rng.random((4, h, w), dtype=np.float32)).to(DEV)
    n = 4 * h * w
    q = np.ascontiguousarray(P.QUANTS)
    qp = C.c_void_p(q.ctypes.data)
    ws = P._nle_workspace(n, lap.device)
    L.check(lib.yond_nle_stats_f32(L.ptr(lap), L.ptr(mean), n, w, qp, len(q), L.ptr(ws), L.stream()), "stats")
    off = (C.c_int * 5)()
    L.check(lib.yond_nle_state_layout(off), "layout")
    heads = []
    for _ in range(3):
        L.check(lib.yond_nle_threshold_f32(L.ptr(lap), n, qp, len(q), 1, L.ptr(ws), L.stream()), "threshold")
        torch.cuda.synchronize()
        heads.append(ws[:off[2]].cpu().numpy().copy())              # ths + sel
    assert np.array_equal(heads[0], heads[1]) and np.array_equal(heads[0], heads[2])
    ths = heads[0][:8 * len(q)].view(np.float64)
    np.testing.assert_array_equal(ths, np.percentile(lap.cpu().numpy().reshape(-1), q))


@pytest.mark.parametrize("H,W,tile_w,k", [(2, 2, 0, 1), (4, 6, 0, 3), (30, 18, 0, 5), (58, 130, 0, 29), (64, 512, 16, 7), (130, 902, 0, 29),
                                          (34, 2064, 0, 9)])
def test_box_kernels_edge_geometries_vs_oracle(H, W, tile_w, k):
    """The sliding-window box kernels (round 3) on the shapes the big tests do not reach -- a 1 x 1 packed image, windows larger
    than the image (multiple reflections), widths that are no multiple of 4 (scalar stores), run-time window sizes (k != 29),
    SIDD_256-style tiling with narrow tiles, strips whose last chunk is partial, one-row segments -- against the oracle's cv2.blur
    restatement: mean / var / lap for self and collab, <= 1 float32 ulp-sized relative error where the value is not tiny."""
    import yond_oracle as O
    from yond_public_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(H * 131 + W)
    lr = rng.random((H, W), dtype=np.float32)
    hr = (rng.random((H, W), dtype=np.float32) * 0.5 + 0.25).astype(np.float32)
    h, w = H // 2, W // 2
    k2 = k // 3 * 2 + 1
    planes = lambda a: np.ascontiguousarray(np.transpose(a, (2, 0, 1)))
    def tiles(a):                                        # the oracle filters every tile_w-wide block on its own
        if not tile_w:
            return [a]
        return np.split(a, a.shape[1] // tile_w, axis=1)
    def filt(fn, a):
        return np.concatenate([fn(t_) for t_ in tiles(a)], axis=1)
    rl, rh = O.bayer2rggb(lr), O.bayer2rggb(hr)
    want = {
        "mean": filt(lambda a: O.box_blur(a, k), rl), "var": filt(lambda a: O.stdfilt(a, k) ** 2, rl),
        "lap": filt(lambda a: O.stdfilt(O.box_blur(a, k2), k), rl),
        "c.mean": filt(lambda a: O.box_blur(a, k), rh), "c.lap": filt(lambda a: O.stdfilt(a, k), rh),
    }
    want["c.var"] = filt(lambda a: O.stdfilt(a, k) ** 2, rl) - want["c.lap"] ** 2
    t, c = torch.from_numpy(lr).to(DEV), torch.from_numpy(hr).to(DEV)
    o = [torch.full((4, h, w), float('nan'), device=DEV) for _ in range(7)]
    st = L.stream()
    L.check(lib.yond_box_stats_self1_f32(L.ptr(t), H, W, k, k2, tile_w, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), st), "self1")
    L.check(lib.yond_box_stats_self2_f32(L.ptr(o[2]), h, w, k, tile_w, L.ptr(o[3]), st), "self2")
    L.check(lib.yond_box_stats_collab_f32(L.ptr(t), L.ptr(c), H, W, k, tile_w, L.ptr(o[4]), L.ptr(o[5]), L.ptr(o[6]), st), "collab")
    torch.cuda.synchronize()
    got = {"mean": o[0], "var": o[1], "lap": o[3], "c.mean": o[4], "c.var": o[5], "c.lap": o[6]}
    for name, g_ in got.items():
        g_ = g_.cpu().numpy()
        ref = planes(want[name]).astype(np.float64)
        assert not np.isnan(g_).any(), (name, "unwritten outputs")
        scale = max(float(np.abs(ref).max()), 1e-30)
        # var / lap are differences of nearly equal float32 numbers: an ulp of E[x^2] is the unit of their error
        unit = 2.0 ** -23 * (1.0 if name.endswith("mean") else float(max(np.abs(planes(want["mean"])).max() ** 2, 1e-30)))
        err = float(np.abs(g_.astype(np.float64) - ref).max())
        assert err <= 4 * max(unit, 2.0 ** -23 * scale) if "lap" not in name else err <= 2e-3 * scale + 1e-6, (name, err, scale)
