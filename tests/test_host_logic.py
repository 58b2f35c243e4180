"""CPU: host-side logic of the product (no kernels): threshold score, moment fit, LUT knot grid, padding,
image sharding -- against the oracle's restatements of the same reference lines."""
import numpy as np
import pytest

import yond_oracle as O
from yond_public_amd import pipeline as P
from yond_public_amd import distributed as D


def test_get_p2d_matches_reference_formula():
    for shape in [(1, 4, 1500, 2000), (1, 4, 128, 128), (1, 4, 60, 68), (2, 4, 33, 95)]:
        assert P.get_p2d(shape, 32) == O.get_p2d(shape, 32)
    assert P.get_p2d((1, 4, 1500, 2000), 32) == (8, 8, 2, 2)


def test_bias_knots_bit_identical():
    for mx in (959.7, 312.4, 41.3, 50.0, 499.2, 500.0, 3.2):
        ub = np.ceil(np.float32(mx)) + 1
        a, b = P._bias_knots(ub), O.bias_knots(ub)
        assert a.dtype == b.dtype and np.array_equal(a, b)


def test_score3_and_fit_from_moments():
    rng = np.random.default_rng(0)
    n = 50000
    lap = (rng.random(n).astype(np.float32) ** 2) * 0.05
    mean = rng.random(n).astype(np.float32)
    var = (0.004 * mean + 5e-5 + 1e-5 * rng.standard_normal(n)).astype(np.float32)
    th, pct, info = O.get_threshold_score3(lap, mean, step=5, full=True)
    quants = np.linspace(5, 100, 20)
    ths = info['ths']
    # emulate what the accumulate kernel returns
    occ = np.zeros((20, P.NBINS // 32), np.uint32)      # bitmap words, as the kernel writes them
    mom = np.zeros((21, 2, 5))
    i_le = np.searchsorted(ths, lap.astype(np.float64), side='left')
    i_lt = np.searchsorted(ths, lap.astype(np.float64), side='right')
    bins = (mean.clip(0, 1) * 1000).astype(int)
    for i, b in zip(i_le, bins):
        if i < 20:
            occ[i, b >> 5] |= np.uint32(1) << np.uint32(b & 31)
    ns = (mean > np.float32(1e-4)) & (mean < np.float32(0.8))
    for k in range(21):
        for j, sel in enumerate((i_lt == k, (i_lt == k) & ns)):
            m, v = mean[sel].astype(np.float64), var[sel].astype(np.float64)
            mom[k, j] = [m.size, m.sum(), v.sum(), (m * m).sum(), (m * v).sum()]
    th2, pct2, info2 = P._score3(ths, quants, occ.view(np.int32))
    assert th2 == th and pct2 == pct
    np.testing.assert_array_equal(info2['npeaks'], info['npeaks'])
    sel = mom[:info2['index'] + 1].sum(axis=0)
    reg = P._fit_from_moments(sel[0], sel[1])
    ref = O.polyfit(mean[lap < th], var[lap < th])
    np.testing.assert_allclose(reg, ref, rtol=1e-8)


def test_vst_scalar_matches_oracle():
    for K, s in [(0.72, 1.8), (4.37, 6.27), (22.65, 37.09)]:
        for x in (0, 959.0, 12.5):
            assert P.vst_scalar(x, np.float64(s), np.float64(K)) == O.VST(x, np.float64(s), gain=np.float64(K))


def test_shard_indices():
    assert D.shard_indices(40, 3, 8) == [3, 11, 19, 27, 35]
    allidx = sorted(i for r in range(8) for i in D.shard_indices(40, r, 8))
    assert allidx == list(range(40))
    sizes = [3000 * 5328] * 8 + [2000 * 3000] * 8 + [4000 * 3000] * 8 + [1000 * 1000] * 16
    parts = [D.shard_indices(40, r, 8, sizes) for r in range(8)]
    assert sorted(i for p in parts for i in p) == list(range(40))
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert max(loads) / (sum(loads) / 8) < 1.2


def test_metric_sums_single_process():
    m = D.MetricSums(2)
    m.update([50.0, 51.0], [0.98, 0.99])
    m.update([48.0], [0.97])                       # round 2 ended by the guard: -1 in iter1's meter, 'last' = iter 0's value
    r = m.reduce()
    assert r['count'] == 2 and r['psnr_iter0'] == 49.0 and r['psnr_iter1'] == 25.0
    assert abs(r['ssim_last'] - 0.98) < 1e-12 and r['psnr_last'] == 49.5 and abs(r['ssim_iter1'] + 0.005) < 1e-12


def test_fit_from_moments_rank_deficient_is_minimum_norm():
    """A constant mean map (lens cap, saturated frame) makes the 2x2 normal equations singular.  The reference's
    scipy.linalg.lstsq (utils/isp_algos.py:364) does not raise there (what it returns depends on how LAPACK rounds the
    vanishing singular value); the product returns the minimum-norm least-squares solution instead of dividing by 0."""
    m = np.full(1000, 0.25)
    v = 1e-4 + 1e-5 * np.random.default_rng(0).standard_normal(1000)
    ref = np.linalg.pinv(np.vstack([m, np.ones(len(m))]).T, rcond=1e-10) @ v
    mom = np.array([m.size, m.sum(), v.sum(), (m * m).sum(), (m * v).sum()])
    np.testing.assert_allclose(P._fit_from_moments(mom, mom), ref, rtol=1e-9)
    one = np.array([1.0, 0.3, 2e-4, 0.09, 6e-5])
    ref1 = np.linalg.pinv(np.array([[0.3, 1.0]])) @ np.array([2e-4])
    np.testing.assert_allclose(P._fit_from_moments(one, one), ref1, rtol=1e-9)
    assert np.all(P._fit_from_moments(np.zeros(5), np.zeros(5)) == 0)


def test_stream_applies_and_frame_items():
    """Host logic of the stream drivers (round 6): which configurations pipeline.denoise_stream runs on its device-chain drivers -- the ones whose
    `frames` may carry per-frame parameter dicts -- and how an element of `frames` is read (no GPU: CPU tensors without a device are refused)."""
    import pytest
    import torch
    from yond_public_amd import pipeline as P
    from yond_public_amd import _lib as L
    once = {'k': 29, 'bias_corr': 'pre', 'iter': 'once', 'full_dn': True}
    assert P.stream_applies(once) and P.stream_applies(dict(once, iter='iter', max_iter=1))
    assert not P.stream_applies(dict(once, iter='iter', max_iter=2))                  # only the shipped single re-estimation is streamed
    assert not P.stream_applies(dict(once, full_dn=False))                            # block-wise denoising: the SIDD group driver's job
    assert not P.stream_applies(dict(once, bias_corr='post'))
    assert not P.stream_applies(dict(once, est_type='ours'))
    assert not P.stream_applies(dict(once, cal_est='foi'))
    assert not P.stream_applies(once, p={'rot_cfa': 1})
    assert not P.stream_applies(once, biaslut=object())                               # the 2-D LUT goes through IterDenoise
    p0 = {'wp': 1023, 'bl': 64}
    with pytest.raises(L.YondHipError):
        P._frame_item((torch.zeros(4, 4), {'rot_cfa': 2}), p0, None)
    with pytest.raises(L.YondHipError):
        P._frame_item(torch.zeros(4, 4), p0, None)                                    # a CPU tensor and no device: no CPU fallback
    assert P.STREAM_LANES == 2 and not P.LANE_PIPELINES and not P.LANE_PIPELINES_ONCE
