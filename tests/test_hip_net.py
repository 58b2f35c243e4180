"""GPU: full denoiser forwards (archs plugin surface, NCHW in/out) against the reference's own outputs
(tests/golden/net.npz, produced by running the reference) and the oracle.
Tolerance: float32 nets, max |delta| <= 1e-4 on O(1) outputs."""
import numpy as np
import pytest
import torch

from hip_common import ARCHS, make_net, report

DEV = "cuda:0"

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


@pytest.mark.parametrize("aname,shape", [("gru32", (1, 4, 160, 224)), ("snr32", (2, 4, 96, 128)), ("gru32", (1, 4, 416, 672))])
def test_net_fp16_path_split_plane_flow_equals_plain_tensors(aname, shape):
    """The fp16 path's forward in the split-plane data flow on H-ONLY planes (engine.HALF_FLOW; round 6) against the same path on plain
    float32 tensors: every operand is the same half-rounded value either way (the plain kernels round when they stage, the flow's producers
    store the rounded value), so the two forwards differ only by the summation order of the layers that changed kernels (stride 2, decoder
    GEMM: generic fp16 kernel -> split family) -- >= 70 dB on the [0, 1] output, and both >= 55 dB from the float32 reference."""
    import yond_oracle as O
    from yond_public_amd import engine as E
    arch = dict(ARCHS[aname])
    net, sd = make_net(arch, 9)
    x = torch.rand(shape, generator=torch.Generator().manual_seed(5)) * 0.9
    t = torch.full((shape[0], 1, 1, 1), 0.05)
    torch.set_num_threads(8)
    ref = O.net_forward(arch, sd, x, t).numpy().astype(np.float64)
    net.precision = 'fp16'
    plan = net._get_plan(torch.device('cuda:0'))
    assert E.HALF_FLOW and plan._sp_flow(shape[0], shape[2], shape[3])
    with torch.no_grad():
        plan.prof = []
        y_flow = net(x.to('cuda:0'), t.to('cuda:0')).cpu().numpy().astype(np.float64)
        tags, plan.prof = [p[0] for p in plan.prof], None
        # every MFMA convolution of the forward is an h-only launch of the split family (no generic kernel, no conversion inside a kernel)
        assert len(tags) == 26 and all(tg.startswith("conv_split_kernel<") and tg.endswith(",1>") for tg in tags), tags
        E.HALF_FLOW = False
        try:
            assert not plan._sp_flow(shape[0], shape[2], shape[3])
            y_plain = net(x.to('cuda:0'), t.to('cuda:0')).cpu().numpy().astype(np.float64)
        finally:
            E.HALF_FLOW = True
    net.precision = 'fp32'
    psnr = lambda a, b: 10 * np.log10(1.0 / max(float(np.mean((a - b) ** 2)), 1e-30))
    print(f"[parity] {aname} {shape}: fp16 path, h-only flow vs plain tensors {psnr(y_flow, y_plain):.1f} dB; vs fp32 reference: flow {psnr(y_flow, ref):.1f} dB, "
          f"plain {psnr(y_plain, ref):.1f} dB")
    assert np.isfinite(y_flow).all()
    assert psnr(y_flow, y_plain) >= 70.0
    assert psnr(y_flow, ref) >= 55.0 and psnr(y_plain, ref) >= 55.0
    assert psnr(y_flow, ref) >= psnr(y_plain, ref) - 1.0


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


def test_split_path_range_guard():
    """The split-operand / fp16 paths stage activations as fp16: |a| > 65504 would become inf silently.  A layer whose
    output is ~1e6 (and whose consumer scales it back) must give the strict fp32 result, with a warning -- and weights
    outside fp16's range must keep their layer off the half-precision kernels."""
    import warnings
    import yond_oracle as O
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    arch = ARCHS["gru8"]
    sd = O.procedural_state_dict(arch, 9)
    sd['conv1.conv1.weight'] = sd['conv1.conv1.weight'] * 1e6          # conv1's output ~ 1e6 ...
    sd['conv1.conv1.bias'] = sd['conv1.conv1.bias'] * 1e6
    sd['conv1.conv2.weight'] = sd['conv1.conv2.weight'] * 1e-6         # ... conv2 brings it back to O(1)
    net = A.GuidedResUnet(dict(arch))
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    noisy, _ = O.synth_noisy(128, 192, 4.0, 6.0, 77)
    p = O.default_params()
    p['gain'], p['sigma'] = np.float64(4.0), np.float64(6.0)
    torch.set_num_threads(8)
    ref = O.VST_Denoiser(noisy, p, arch, sd, bias_corr='pre')
    assert np.isfinite(ref).all()
    x = torch.from_numpy(noisy).to(DEV)
    plan = P._plan_of(net, x.device)
    assert plan.blocks[1]['conv1'].split(2) is None                    # 1e6-sized weights: refused by the split packing
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        dn = P.VST_Denoiser(x, p, net, arch, bias_corr='pre').cpu().numpy()
    assert any("fp16's range" in str(w.message) for w in wlist)
    assert np.isfinite(dn).all()
    assert report("guarded forward vs oracle", dn, ref) <= 2e-4
    # without the guard the same forward is wrong (this is what the guard prevents)
    raw = P.VST_Denoiser(x, p, net, arch, bias_corr='pre', guard=False).cpu().numpy()
    assert not np.isfinite(raw).all() or np.abs(raw - ref).max() > 1e-2
    # the stream driver recomputes the offending frame, too
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'full_dn': True}
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        outs = list(P.denoise_stream([x, x.clone()], net, arch, pipe))
    assert len(outs) == 2 and all(bool(torch.isfinite(o['raw_dns'][0]).all()) for o in outs)
    assert sum("fp16's range" in str(w.message) for w in wlist) == 2
    # the archs plugin surface `net(x, t)` (engine.DenoiserPlan.forward_nchw) is guarded as well: same network, NCHW input
    xn = torch.rand((1, 4, 64, 96), generator=torch.Generator().manual_seed(3))
    tn = torch.tensor(0.04)
    refn = O.net_forward(arch, sd, xn, tn).numpy()
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        with torch.no_grad():
            yn = net(xn.to(DEV), tn.to(DEV)).cpu().numpy()
    assert any("fp16's range" in str(w.message) for w in wlist)
    assert np.isfinite(yn).all() and report("guarded net(x, t) vs oracle", yn, refn) <= 2e-4 * max(1.0, float(np.abs(refn).max()))
    # a well-scaled network does not trip it
    net2, _ = make_net(arch, 9)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        P.VST_Denoiser(x, p, net2, arch, bias_corr='pre')
    assert not any("fp16's range" in str(w.message) for w in wlist)


def test_engine_flow_switches_are_bit_identical():
    """The data-flow variants of one GuidedResUnet forward -- every tensor [N][H][W][C]; split planes; + alternating tile order, two
    sub-positions per tile in the last decoder GEMM, the stride-2 layers' second output (conv1 by LDS-DMA at the deep levels) --
    give the same bits: they move the same float32 values through different layouts and orders."""
    import torch
    from yond_public_amd import engine as E
    from yond_public_amd import pipeline as P
    arch = ARCHS["gru32"]
    net, _ = make_net(arch, 5)
    plan = P._plan_of(net, torch.device(DEV))
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.rand((2, 160, 224, 4), device=DEV, generator=g)
    t = torch.tensor([0.03, 0.08], device=DEV)
    ub = x.reshape(2, -1).max(1).values.contiguous()
    saved = (E.SPLIT_PLANES, E.SP_FLOW, E.SNAKE_ORDER, E.K1_SUB2, E.SP_CONV1_MIN_LEVEL)
    outs = []
    try:
        for cfg in ((False, False, False, False, 99), (True, False, False, False, 99), (True, True, False, False, 99),
                    (True, True, True, False, 99), (True, True, True, True, 99), (True, True, True, True, 3), (True, True, True, True, 2)):
            E.SPLIT_PLANES, E.SP_FLOW, E.SNAKE_ORDER, E.K1_SUB2, E.SP_CONV1_MIN_LEVEL = cfg
            outs.append(plan.forward_nhwc4(x, t, ub=ub).clone())
    finally:
        E.SPLIT_PLANES, E.SP_FLOW, E.SNAKE_ORDER, E.K1_SUB2, E.SP_CONV1_MIN_LEVEL = saved
    torch.cuda.synchronize()
    assert bool(torch.isfinite(outs[0]).all())
    for i, o in enumerate(outs[1:], 1):
        assert torch.equal(o, outs[0]), (i, float((o - outs[0]).abs().max()))


def test_unet_split_plane_pairs_bit_identical():
    """UNetSeeInDark with the stage-internal tensors in split planes (engine.UNET_SP) against every tensor [N][H][W][C]: same bits."""
    import torch
    from yond_public_amd import engine as E
    from yond_public_amd import pipeline as P
    arch = ARCHS["unet32"]
    net, _ = make_net(arch, 3)
    plan = P._plan_of(net, torch.device(DEV))
    g = torch.Generator(device=DEV).manual_seed(10)
    x = torch.rand((2, 96, 160, 4), device=DEV, generator=g)
    ub = x.reshape(2, -1).max(1).values.contiguous()
    saved = E.UNET_SP
    try:
        E.UNET_SP = False
        a = plan.forward_nhwc4(x, None, ub=ub).clone()
        E.UNET_SP = True
        b = plan.forward_nhwc4(x, None, ub=ub).clone()
    finally:
        E.UNET_SP = saved
    torch.cuda.synchronize()
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b), float((a - b).abs().max())


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_two_lanes_forward_concurrently_bit_equal(precision):
    """Round 6: the stream drivers run the network passes of consecutive frames on two HIP streams, each engine `lane` with its own split-plane
    tensors and FiLM vectors (DenoiserPlan.lane).  Two DIFFERENT inputs with different noise levels queued on the two lanes without any
    synchronisation between them must each come out bit for bit as when forwarded alone -- several times over, so that the lanes' persistent
    tensors (whose zero padding is written once) are reused."""
    from yond_public_amd import pipeline as P
    arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
    net, _ = make_net(arch, 9)
    net.precision = precision
    plan = P._plan_of(net, torch.device(DEV))
    g = torch.Generator().manual_seed(5)
    xa = torch.rand((1, 96, 160, 4), generator=g).to(DEV)
    xb = (torch.rand((1, 96, 160, 4), generator=g) * 0.5).to(DEV)
    ta, tb = torch.tensor([0.02], device=DEV), torch.tensor([0.07], device=DEV)
    alone_a = plan.forward_nhwc4(xa, ta).clone()
    alone_b = plan.forward_nhwc4(xb, tb).clone()
    torch.cuda.synchronize()
    assert float((alone_a - alone_b).abs().max()) > 1e-3
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    for rep in range(3):
        side.wait_stream(main)
        plan.lane = 1
        with torch.cuda.stream(side):
            yb = plan.forward_nhwc4(xb, tb)
        plan.lane = 0
        ya = plan.forward_nhwc4(xa, ta)
        main.wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(ya, alone_a) and torch.equal(yb, alone_b), f"repetition {rep}"
    assert plan.lane == 0
