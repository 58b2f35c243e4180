"""GPU: IterDenoise (row Q) against the reference's captured outputs, metrics kernel, full-frame smoke."""
import numpy as np
import pytest
import torch

from hip_common import ARCHS, make_net, report, sha

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize("ci", range(2))
def test_iter_denoise_matches_reference(golden, ci):
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    g = golden("iter")
    noisy, clean = O.synth_noisy(256, 8192, 4.0, 6.0, 31)
    full, _ = O.synth_noisy(512, 1024, 4.0, 6.0, 32)
    assert np.array_equal(sha(noisy), g["sha_noisy"]) and np.array_equal(sha(full), g["sha_full"])
    arch = ARCHS[str(g[f"arch_{ci}"])]
    net, sd = make_net(arch, int(g[f"seed_{ci}"]))
    full_dn = bool(g[f"full_dn_{ci}"])
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': full_dn,
            'collab_sidd256': True}
    lr = noisy if full_dn else np.array(np.split(noisy, 32, axis=-1))
    res = P.IterDenoise(lr, net, arch, pipe, lr_full=full, device=DEV)
    regs = g[f"regs_{ci}"]
    assert len(res['regs']) == len(regs)
    for r, gr in zip(res['regs'], regs):
        print(f"[parity] regs {r[0]:.6e},{r[1]:.6e} vs {gr[0]:.6e},{gr[1]:.6e}")
        np.testing.assert_allclose(r[0], gr[0], rtol=2e-5)
        np.testing.assert_allclose(r[1], gr[1], rtol=0, atol=2e-5 * abs(gr[0]) + 1e-9)
    for it, dn in enumerate(res['raw_dns']):
        dn = dn.cpu().numpy()
        assert report(f"IterDenoise case {ci} iter {it}", dn[:, :768], g[f"dn_{ci}_{it}_crop"]) <= 1e-4
        np.testing.assert_allclose(np.asarray(dn, np.float64).sum(), g[f"dn_{ci}_{it}_chk"][0], rtol=1e-5)


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
