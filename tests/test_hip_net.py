"""GPU: full denoiser forwards (archs plugin surface, NCHW in/out) against the reference's own outputs
(tests/golden/net.npz, produced by running the reference) and the oracle.
Tolerance: float32 nets, max |delta| <= 1e-4 on O(1) outputs."""
import numpy as np
import pytest
import torch

from hip_common import ARCHS, make_net, report

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ci", range(7))
def test_net_forward_matches_reference(golden, ci):
    g = golden("net")
    aname = str(g[f"arch_{ci}"])
    arch = ARCHS[aname]
    meta = g[f"meta_{ci}"]
    shape = tuple(int(v) for v in meta[2:])
    net, sd = make_net(arch, int(meta[0]))
    x = (torch.rand(shape, generator=torch.Generator().manual_seed(int(meta[1]))) * 0.9).to('cuda:0')
    with torch.no_grad():
        if f"t_{ci}" in g.files:
            t = torch.from_numpy(g[f"t_{ci}"]).to('cuda:0')
            y = net(x, t)
        else:
            y = net(x)
    torch.cuda.synchronize()
    ref = g[f"y_{ci}"]
    err = report(f"net {aname} {shape}", y.cpu().numpy(), ref)
    assert err <= 1e-4 * max(1.0, float(np.abs(ref).max()))


def test_net_larger_frame_vs_oracle():
    import yond_oracle as O
    arch = ARCHS["gru32"]
    net, sd = make_net(arch, 5)
    x = torch.rand((1, 4, 160, 224), generator=torch.Generator().manual_seed(77))
    t = torch.tensor(0.04)
    torch.set_num_threads(8)
    ref = O.net_forward(arch, sd, x, t).numpy()
    with torch.no_grad():
        y = net(x.to('cuda:0'), t.to('cuda:0'))
    err = report("net gru32 160x224", y.cpu().numpy(), ref)
    assert err <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    # state_dict reload invalidates the cached plan
    sd2 = O.procedural_state_dict(arch, 6)
    net.load_state_dict(sd2)
    ref2 = O.net_forward(arch, sd2, x, t).numpy()
    with torch.no_grad():
        y2 = net(x.to('cuda:0'), t.to('cuda:0'))
    assert report("net gru32 reloaded", y2.cpu().numpy(), ref2) <= 1e-4 * max(1.0, float(np.abs(ref2).max()))


@pytest.mark.parametrize("aname", ["gru32", "snr32", "unet32"])
def test_net_fp16_mfma_path_vs_fp32(aname):
    """BASELINE cfg 5: convolutions on the fp16 MFMA path (operands rounded to half at the matrix core, fp32
    accumulation, fp32 tensors).  SURVEY section 8d: PSNR(fp16 path, fp32 reference) >= 55 dB on the [0, 1] output."""
    import yond_oracle as O
    arch = dict(ARCHS[aname])
    net, sd = make_net(arch, 9)
    x = torch.rand((1, 4, 160, 224), generator=torch.Generator().manual_seed(5)) * 0.9
    t = torch.tensor(0.05)
    guided = 'guided' in arch
    torch.set_num_threads(8)
    ref = (O.net_forward(arch, sd, x, t) if guided else O.net_forward(arch, sd, x)).numpy()
    net.precision = 'fp16'
    with torch.no_grad():
        y = (net(x.to('cuda:0'), t.to('cuda:0')) if guided else net(x.to('cuda:0'))).cpu().numpy()
    mse = float(np.mean((y.astype(np.float64) - ref.astype(np.float64)) ** 2))
    psnr = 10 * np.log10(1.0 / mse)
    print(f"[parity] {aname} fp16-MFMA path vs fp32 reference: PSNR {psnr:.1f} dB, max |delta| {np.abs(y - ref).max():.3e}")
    assert psnr >= 55.0
    net.precision = 'fp32'                       # the plan key holds the precision: back to the fp32 kernels
    with torch.no_grad():
        y32 = (net(x.to('cuda:0'), t.to('cuda:0')) if guided else net(x.to('cuda:0'))).cpu().numpy()
    assert report(f"net {aname} back on fp32", y32, ref) <= 1e-4 * max(1.0, float(np.abs(ref).max()))
