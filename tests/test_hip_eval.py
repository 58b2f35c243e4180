"""GPU: the evaluation entry point with the reference's surface (yond_public_amd/YOND_SIDD.py, mirroring
YOND_SIDD.py:136-236, 485-570), the batched full-frame driver (BASELINE cfg 4) and the RCCL metric reduction."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from hip_common import ARCHS, report

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNFILE = os.path.join(ROOT, "runfiles", "YOND", "SIDD_simple+full_pre_grumix.yml")


def test_yond_sidd_eval_synthetic(tmp_path, monkeypatch):
    """`YOND_SIDD.py -f runfile -m eval` on two synthetic SIDD stand-ins (no dataset in the image): both rounds run, the
    per-image PSNR / SSIM the driver logs equal the oracle's block metrics of the driver's own outputs, the outputs and
    estimates equal the oracle's IterDenoise, and the reduced means follow the reference's meter rules (:643-672)."""
    import yond_oracle as O
    from yond_public_amd import YOND_SIDD as Y
    monkeypatch.chdir(tmp_path)                                    # the driver writes ./logs like the reference
    trainer = Y.YOND_SIDD(['-f', RUNFILE, '-m', 'eval', '--synthetic', '2'])
    red = trainer.eval(-1)
    assert red['count'] == 2
    arch = dict(trainer.arch)
    sd = O.denoising_state_dict(arch, 0)
    torch.set_num_threads(8)
    p0, s0, p1, s1 = [], [], [], []
    for k in range(2):
        data = trainer.dst_eval[k]
        m = trainer.metrics[data['name']]
        assert len(m['psnr']) == 2 and len(m['reg']) == 2         # round 2 ran
        res = trainer.IterDenoise(data, {'p': dict(trainer.pipe, wp=1023, bl=64, ratio=1, gain=1, sigma=0, scale=959.0), 'img_id': k})
        hr = np.concatenate(data['hr'], axis=-1)
        for it in range(2):
            dn = res['raw_dns'][it].cpu().numpy()
            ps, ss = O.sidd_block_metrics(dn, hr)
            assert abs(ps - m['psnr'][it]) < 1e-5 and abs(ss - m['ssim'][it]) < 1e-7
        if k == 0:                                                 # the whole pipeline against the oracle for one image
            ref = O.IterDenoise(data['lr'], arch, sd, dict(trainer.pipe), lr_full=data['lr_full'])
            assert len(ref['raw_dns']) == 2
            for it in range(2):
                assert report(f"eval image 0 iter {it}", res['raw_dns'][it].cpu().numpy(), ref['raw_dns'][it]) <= 1e-4
                np.testing.assert_allclose(res['regs'][it][0], ref['regs'][it][0], rtol=2e-5)
        p0.append(m['psnr'][0]); s0.append(m['ssim'][0]); p1.append(m['psnr'][1]); s1.append(m['ssim'][1])
    assert abs(red['psnr_iter0'] - np.mean(p0)) < 1e-9 and abs(red['ssim_iter1'] - np.mean(s1)) < 1e-12
    assert abs(red['psnr_last'] - np.mean(p1)) < 1e-9 and abs(red['ssim_last'] - np.mean(s1)) < 1e-12


def test_grouped_images_equal_the_per_image_run(tmp_path, monkeypatch):
    """YOND_SIDD.eval denoises `--group` images together (round 1 of the group = ONE batch-(32 G) forward, round 2 another).
    (1) The network itself: a batch-96 forward (three images' blocks, three different t) equals the three batch-32 forwards BIT FOR BIT --
    other tile shapes and folded tiles than a batch of 32 takes, the same arithmetic per pixel.  (2) Per image the outputs of both rounds,
    the estimates and the logged metrics are those of the one-image IterDenoise to what two IterDenoise runs of one image differ by (the
    estimator's float64 moment sums are accumulated with atomics: 1e-12 relative on the estimates, <= 5e-6 on the outputs) -- three images of
    different noise levels, and the driver's `--group 3` against `--group 1` on five images (groups of 3 + 2)."""
    from yond_public_amd import YOND_SIDD as Y
    from yond_public_amd import pipeline as P
    monkeypatch.chdir(tmp_path)
    trainer = Y.YOND_SIDD(['-f', RUNFILE, '-m', 'eval', '--synthetic', '5', '--group', '3'])
    # (1) the forward
    plan = P._plan_of(trainer.net, torch.device(DEV))
    g = torch.Generator().manual_seed(5)
    x = torch.rand(96, 128, 128, 4, generator=g).to(DEV)
    t = torch.tensor([0.02] * 32 + [0.05] * 32 + [0.11] * 32, device=DEV)
    ub = x.reshape(96, -1).max(1).values.contiguous()
    y96 = plan.forward_nhwc4(x, t, ub=ub)
    for gi in range(3):
        sl = slice(32 * gi, 32 * gi + 32)
        y32 = plan.forward_nhwc4(x[sl].contiguous(), t[sl].contiguous(), ub=ub[sl].contiguous())
        assert torch.equal(y96[sl], y32), f"image {gi}: the grouped forward differs from the batch-32 forward"
    # (2) the pipeline
    items = []
    for j, (K, sg) in enumerate([(4.0, 6.0), (1.5, 3.0), (9.0, 14.0)]):
        ds = Y.SyntheticSIDD(1, K=K, sigma=sg, full_hw=(1024, 1536))
        d = ds[0]
        items.append({k: (torch.from_numpy(np.ascontiguousarray(v)).to(DEV) if isinstance(v, np.ndarray) else v) for k, v in d.items()})
    p = dict(trainer.pipe, wp=1023, bl=64, ratio=1, gain=1, sigma=0, scale=959.0)
    singles = [trainer.IterDenoise(d, {'p': dict(p), 'img_id': i}) for i, d in enumerate(items)]
    again = trainer.IterDenoise(items[0], {'p': dict(p), 'img_id': 0})
    grouped = trainer.IterDenoiseGroup(items, [{'p': dict(p), 'img_id': i} for i in range(3)])
    assert len({float(r['regs'][0][0]) for r in singles}) == 3                  # three different estimates
    print(f"[parity] two runs of one image differ by {float((again['raw_dns'][1] - singles[0]['raw_dns'][1]).abs().max()):.2e}")
    for gi, (a, b) in enumerate(zip(singles, grouped)):
        assert len(a['raw_dns']) == len(b['raw_dns']) == 2
        for it, (x_, y_) in enumerate(zip(a['raw_dns'], b['raw_dns'])):
            assert report(f"grouped image {gi} round {it} vs one image at a time", y_.cpu().numpy(), x_.cpu().numpy()) <= 5e-6
        np.testing.assert_allclose(np.asarray(b['regs'], np.float64), np.asarray(a['regs'], np.float64), rtol=1e-9, atol=0)
        np.testing.assert_allclose(np.asarray(b['params'], np.float64), np.asarray(a['params'], np.float64), rtol=1e-9, atol=0)
    # the driver: groups of 3 + 2 against one image at a time
    red3 = trainer.eval(-1)
    m3 = {k: dict(v) for k, v in trainer.metrics.items()}
    trainer.parser.group = 1
    red1 = trainer.eval(-1)
    assert red3['count'] == red1['count'] == 5
    for key in red1:
        assert abs(red1[key] - red3[key]) <= 1e-5 * max(1.0, abs(red1[key])), key
    for k, v in trainer.metrics.items():
        np.testing.assert_allclose(v['psnr'], m3[k]['psnr'], rtol=0, atol=1e-4)
        np.testing.assert_allclose(v['ssim'], m3[k]['ssim'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.asarray(v['reg'], np.float64), np.asarray(m3[k]['reg'], np.float64), rtol=1e-9, atol=0)


def test_streamed_groups_equal_one_group_at_a_time(tmp_path, monkeypatch):
    """pipeline.denoise_stream_groups -- consecutive groups of SIDD items overlapped on two HIP streams (group k+1's full-frame estimates under group k's
    first pass, group k's collaborative estimates under group k-1's second), the `finish` callback (the driver's block metrics) on a third -- returns, image
    by image, what IterDenoiseGroup returns one group at a time: seven images of four noise levels in groups of 3 + 3 + 1 (more groups than the buffer ring holds
    would wrap it: nine groups of one image do), the metrics computed inside the callback equal to those computed afterwards; and YOND_SIDD.eval's
    streamed loop equals its `--no-stream` loop."""
    from yond_public_amd import YOND_SIDD as Y
    from yond_public_amd import pipeline as P
    monkeypatch.chdir(tmp_path)
    trainer = Y.YOND_SIDD(['-f', RUNFILE, '-m', 'eval', '--synthetic', '9', '--group', '4'])
    items = []
    for j, (K, sg) in enumerate([(4.0, 6.0), (1.5, 3.0), (9.0, 14.0), (2.5, 9.0), (4.0, 6.0), (1.5, 3.0), (9.0, 14.0)]):
        d = Y.SyntheticSIDD(1, K=K, sigma=sg, full_hw=(1024, 1536) if j % 2 else (768, 2048))[0]
        items.append({k: (torch.from_numpy(np.ascontiguousarray(v)).to(DEV) if isinstance(v, np.ndarray) else v) for k, v in d.items()})
    pipe = dict(trainer.pipe)
    p = dict(pipe, wp=1023, bl=64, ratio=1, gain=1, sigma=0, scale=959.0)
    pairs = [(d['lr'], d['lr_full']) for d in items]
    hr = [torch.cat(list(d['hr']), dim=-1) for d in items]

    def run(groups):
        seen, out = {}, []

        def finish(gi, ress):
            seen[gi] = [[P.block_metrics(dn, hr[groups[gi][g]]) for dn in r['raw_dns']] for g, r in enumerate(ress)]
        for gi, ress in enumerate(P.denoise_stream_groups(([pairs[i] for i in g] for g in groups), trainer.net, trainer.arch, pipe, p=p, finish=finish)):
            assert len(ress) == len(groups[gi]) and gi in seen
            for g, r in enumerate(ress):
                again = [P.block_metrics(dn, hr[groups[gi][g]]) for dn in r['raw_dns']]
                for (a0, a1), (b0, b1) in zip(seen[gi][g], again):
                    assert np.array_equal(a0, b0) and np.array_equal(a1, b1)
                out.append(r)
        return out
    want = []
    for g in ([0, 1, 2], [3, 4, 5], [6]):
        want += P.IterDenoiseGroup([pairs[i] for i in g], trainer.net, trainer.arch, pipe, ps=p)
    for groups in ([[0, 1, 2], [3, 4, 5], [6]], [[i % 7] for i in range(9)]):
        got = run(groups)
        order = [i for g in groups for i in g]
        assert len(got) == len(order)
        for r, i in zip(got, order):
            w = want[i]
            assert len(r['raw_dns']) == len(w['raw_dns']) == 2
            for it in range(2):
                assert report(f"streamed groups {len(groups)}: image {i} round {it}", r['raw_dns'][it].cpu().numpy(), w['raw_dns'][it].cpu().numpy()) <= 5e-6
            np.testing.assert_allclose(np.asarray(r['regs'], np.float64), np.asarray(w['regs'], np.float64), rtol=1e-9, atol=0)
    # the driver: streamed against one group at a time
    assert P.STREAM_GROUPS and trainer.parser.stream
    red_s = trainer.eval(-1)
    m_s = {k: dict(v) for k, v in trainer.metrics.items()}
    trainer.parser.stream = False
    red_g = trainer.eval(-1)
    assert red_s['count'] == red_g['count'] == 9
    for key in red_g:
        assert abs(red_g[key] - red_s[key]) <= 1e-5 * max(1.0, abs(red_g[key])), key
    for k, v in trainer.metrics.items():
        np.testing.assert_allclose(v['psnr'], m_s[k]['psnr'], rtol=0, atol=1e-4)
        np.testing.assert_allclose(v['ssim'], m_s[k]['ssim'], rtol=0, atol=1e-6)


def test_yond_sidd_full_dn_runfile(tmp_path, monkeypatch):
    """A runfile with full_dn: True (the ELD / LRID / DND style of SURVEY 3.2) must run on the SIDD stack the dataset
    yields: the driver concatenates first, as YOND_SIDD.py:387-389 does."""
    import yaml
    import yond_oracle as O
    from yond_public_amd import YOND_SIDD as Y
    cfg = yaml.load(open(RUNFILE), Loader=yaml.FullLoader)
    cfg['pipeline']['full_dn'] = True
    cfg['arch']['nf'] = 8
    rf = tmp_path / "full_dn.yml"
    rf.write_text(yaml.dump(cfg))
    monkeypatch.chdir(tmp_path)
    trainer = Y.YOND_SIDD(['-f', str(rf), '-m', 'eval', '--synthetic', '1'])
    red = trainer.eval(-1)
    data = trainer.dst_eval[0]
    arch = dict(trainer.arch)
    torch.set_num_threads(8)
    ref = O.IterDenoise(data['lr'], arch, O.denoising_state_dict(arch, 0), dict(trainer.pipe), lr_full=data['lr_full'])
    m = trainer.metrics[data['name']]
    assert len(m['psnr']) == len(ref['raw_dns']) == 2
    hr = np.concatenate(data['hr'], axis=-1)
    for it in range(2):
        ps, ss = O.sidd_block_metrics(ref['raw_dns'][it], hr)
        assert abs(ps - m['psnr'][it]) < 2e-3 and abs(ss - m['ssim'][it]) < 2e-5
        np.testing.assert_allclose(m['reg'][it][0], ref['regs'][it][0], rtol=2e-5)


def test_streamed_batches_equal_iterdenoise():
    """pipeline.denoise_stream_batches (cfg 4's driver: B frames per forward, the estimators of batch k+1 on a second HIP stream under the forward of
    batch k): every frame comes out as IterDenoise returns it -- seven frames in batches of 3 + 3 + 1, UNetSeeInDark and the guided net."""
    import yond_oracle as O
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'once', 'max_iter': 1, 'full_dn': True}
    for aname, seed in (("unet32", 82), ("gru32", 9)):
        arch = ARCHS[aname]
        net = getattr(A, arch['name'])(dict(arch))
        net.load_state_dict(O.denoising_state_dict(arch, seed))
        net = net.to(DEV).eval()
        frames = [torch.from_numpy(O.synth_noisy(192, 320, 2.0 + i, 4.0 + 3 * i, 40 + i)[0]).to(DEV) for i in range(7)]
        want = [P.IterDenoise(f, net, arch, pipe) for f in frames]
        got = list(P.denoise_stream_batches(iter(frames), 3, net, arch, pipe))
        assert len(got) == 7
        for i, (r, w) in enumerate(zip(got, want)):
            assert len(r['raw_dns']) == 1
            assert report(f"{aname}: streamed batches, frame {i}", r['raw_dns'][0].cpu().numpy(), w['raw_dns'][0].cpu().numpy()) <= 5e-6
            np.testing.assert_allclose(np.asarray(r['regs'], np.float64), np.asarray(w['regs'], np.float64), rtol=1e-9, atol=0)
            np.testing.assert_allclose(np.asarray(r['params'], np.float64), np.asarray(w['params'], np.float64), rtol=1e-9, atol=0)


def test_iter_denoise_batch_equals_per_frame():
    """BASELINE cfg 4 driver: B frames with their own estimates through ONE batched forward per round give, frame by
    frame, IterDenoise's result (UNetSeeInDark and the guided net; two rounds)."""
    import yond_oracle as O
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    from yond_public_amd import synthetic as S
    for aname in ("unet8", "gru8"):
        arch = ARCHS[aname]
        net = getattr(A, arch['name'])(dict(arch))
        net.load_state_dict(S.denoising_state_dict(net, 5))
        net = net.to(DEV).eval()
        pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': True}
        frames = [torch.from_numpy(S.synth_noisy(320, 448, 2.0 + i, 14.0 + 3 * i, 60 + i)[0]).to(DEV) for i in range(3)]
        seq = [P.IterDenoise(f, net, arch, pipe) for f in frames]
        bat = P.IterDenoiseBatch(frames, net, arch, pipe)
        assert len(bat['raw_dns']) == 2 and all(bat['alive'])
        for i, s in enumerate(seq):
            assert len(s['raw_dns']) == 2
            for it in range(2):
                assert float((bat['raw_dns'][it][i] - s['raw_dns'][it]).abs().max()) <= 2e-6
                np.testing.assert_allclose(bat['regs'][it][i][0], s['regs'][it][0], rtol=1e-6)


NCCL_WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, os.environ["YOND_ROOT"])
    import torch
    from yond_public_amd import distributed as D
    rank, local, world = D.init()                    # backend "nccl" == RCCL
    import torch.distributed as dist
    assert dist.get_backend() == "nccl" and world == 2
    torch.cuda.set_device(local)
    sums = D.MetricSums(2)
    for k in D.shard_indices(7, rank, world):
        sums.update([40.0 + k, 41.0 + k], [0.9 + 0.01 * k, 0.91 + 0.01 * k])
    D.barrier()
    t = D.max_over_ranks(1.0 + rank, torch.device("cuda", local))
    red = sums.reduce(torch.device("cuda", local))
    if rank == 0:
        print("RESULT " + json.dumps({"red": red, "t": t}))
    dist.destroy_process_group()
''')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL metric reduction)")
def test_two_rank_rccl_metric_reduction(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, YOND_ROOT=ROOT, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert r["t"] == 2.0 and r["red"]["count"] == 7
    assert abs(r["red"]["psnr_last"] - sum(41.0 + k for k in range(7)) / 7) < 1e-12


def _torchrun_one_rank(args, cwd, port):
    """Run a driver as a CHILD job under torchrun with ONE rank, backend "nccl" (= RCCL): the process group, the barrier and
    the all-reduces are the very calls an eight-GPU node executes.  (A child: never re-exec a process that touched the GPU.)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    out = subprocess.run(cmd, env=env, cwd=str(cwd), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return out


def test_bench_one_rank_under_torchrun_runs_rccl(tmp_path):
    """bench.py as the driver launches it for N > 1, with N = 1: a process group on backend nccl, barrier + max-over-ranks +
    the PSNR all-reduce really issued (counted), one JSON line with n_gpus 1."""
    out = _torchrun_one_rank([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--min-warmup-s", "0",
                              "--frames-per-step", "2", "--no-extras", "--no-cpu-baseline"], tmp_path, 29541)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["value"] > 0
    assert r["collectives"]["backend"] == "nccl"
    assert r["collectives"]["all_reduce"] >= 2 and r["collectives"]["barrier"] >= 2


def test_yond_sidd_one_rank_under_torchrun_runs_rccl(tmp_path):
    """`YOND_SIDD.py -m eval --synthetic 2` under torchrun --nproc-per-node 1: the metric reduction is an RCCL all-reduce."""
    out = _torchrun_one_rank([os.path.join(ROOT, "YOND_SIDD.py"), "-f", RUNFILE, "-m", "eval", "--synthetic", "2"], tmp_path, 29542)
    text = out.stdout
    assert "2 images on 1 GPU(s)" in text, text[-2000:]
    line = [l for l in text.splitlines() if "collectives:" in l][-1]
    assert "backend=nccl" in line
    assert int(line.split("all_reduce=")[1].split(",")[0]) >= 2


def test_rot90_kernel_and_rot_cfa_pipeline(golden):
    """N3: the rot90 copy kernel (bit exact vs np.rot90) and IterDenoise with p['rot_cfa'] (YOND_SIDD.py:402-404, 462-464)
    against the reference's own run on a GBRG image (tests/golden/rot.npz)."""
    from test_oracle_golden import rot_case, iter_crop
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    from yond_public_amd.utils.sidd_utils import rot_bayer
    rng = np.random.default_rng(1)
    x = rng.random((3, 10, 14)).astype(np.float32)
    for k in range(4):
        assert np.array_equal(P.rot90(torch.from_numpy(x).to(DEV), k).cpu().numpy(), np.rot90(x, k, axes=(-2, -1)))
    t = torch.from_numpy(x[0]).to(DEV)
    assert np.array_equal(rot_bayer(t, [[2, 3], [1, 2]]).cpu().numpy(), np.rot90(x[0], 1))
    g = golden("rot")
    lr, full, arch, sd, pipe, p = rot_case()
    net = A.GuidedResUnet(dict(arch))
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    res = P.IterDenoise(lr, net, arch, pipe, lr_full=full, p=p, device=DEV)
    assert len(res['raw_dns']) == int(g["nout"]) == 2
    for r, gr in zip(res['regs'], g["regs"]):
        np.testing.assert_allclose(r[0], gr[0], rtol=2e-5)
    for it, dn in enumerate(res['raw_dns']):
        for got, tag in zip(iter_crop(dn.cpu().numpy()), ("blk", "seam", "sub")):
            assert report(f"rot_cfa IterDenoise iter {it} {tag}", got, g[f"dn_{it}_{tag}"]) <= 1e-4


def test_eval_on_a_mat_tree_in_the_reference_layout(tmp_path, monkeypatch):
    """`YOND_SIDD.py -m eval` on SIDD_Validation_Raw/*.mat (MATLAB v5, written here) + SIDD_Benchmark_Data metadata: the
    loader feeds the driver, the CFA of the metadata reaches p['cfa'], metrics come out per image."""
    import scipy.io as sio
    import yaml
    import yond_oracle as O
    from yond_public_amd import YOND_SIDD as Y
    root = tmp_path / "SIDD"
    vr = root / "SIDD_Validation_Raw"
    vr.mkdir(parents=True)
    lrs, hrs = [], []
    for i in range(2):
        noisy, clean = O.synth_noisy(256, 8192, 3.0 + i, 12.0, 300 + i)
        lrs.append(np.array(np.split(noisy, 32, axis=-1)))
        hrs.append(np.array(np.split(clean, 32, axis=-1)))
    sio.savemat(vr / "ValidationNoisyBlocksRaw.mat", {"ValidationNoisyBlocksRaw": np.array(lrs)})
    sio.savemat(vr / "ValidationGtBlocksRaw.mat", {"ValidationGtBlocksRaw": np.array(hrs)})
    cfg = yaml.load(open(RUNFILE), Loader=yaml.FullLoader)
    for key in ('dst', 'dst_eval', 'dst_test'):
        cfg[key]['root_dir'] = str(root)
    cfg['arch']['nf'] = 8
    rf = tmp_path / "mat.yml"
    rf.write_text(yaml.dump(cfg))
    monkeypatch.chdir(tmp_path)
    trainer = Y.YOND_SIDD(['-f', str(rf), '-m', 'eval'])
    assert type(trainer.dst_eval).__name__ == 'SIDD_Dataset' and len(trainer.dst_eval) == 2
    red = trainer.eval(-1)
    assert red['count'] == 2 and red['psnr_iter0'] > 20 and len(trainer.metrics) == 2
    arch = dict(trainer.arch)
    torch.set_num_threads(8)
    ref = O.IterDenoise(lrs[0], arch, O.denoising_state_dict(arch, 0), dict(trainer.pipe))
    m = trainer.metrics['sidd_0000']
    for it in range(len(ref['raw_dns'])):
        ps, ss = O.sidd_block_metrics(ref['raw_dns'][it], np.concatenate(hrs[0], axis=-1))
        assert abs(ps - m['psnr'][it]) < 2e-3 and abs(ss - m['ssim'][it]) < 2e-5


def test_iter_denoise_without_estimate_branch(golden):
    """YOND_SIDD.py:358-381 (full_est False): every block through Simple_Denoiser, regs = (0, 0); against the reference's run."""
    import yond_oracle as O
    from test_oracle_golden import iter_crop
    from yond_public_amd import archs as A
    from yond_public_amd import pipeline as P
    g = golden("rot")
    noisy, _ = O.synth_noisy(256, 8192, 2.0, 20.0, 31)
    arch = ARCHS["unet8"]
    net = A.UNetSeeInDark(dict(arch))
    net.load_state_dict(O.denoising_state_dict(arch, 82))
    net = net.to(DEV).eval()
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': 'iter', 'max_iter': 1, 'full_dn': False, 'full_est': False,
            'est_type': 'simple'}
    res = P.IterDenoise(np.array(np.split(noisy, 32, axis=-1)), net, arch, pipe, device=DEV)
    assert res['regs'] == (0, 0) and len(res['raw_dns']) == 1
    for got, tag in zip(iter_crop(res['raw_dns'][0].cpu().numpy()), ("blk", "seam", "sub")):
        assert report(f"Simple_Denoiser branch {tag}", got, g[f"simple_{tag}"]) <= 1e-4


def _small_full_runfile(tmp_path, src, root_dir, nf=8, **dst_over):
    """A copy of a full-frame runfile with a small network and the dataset root pointed at a miniature tree."""
    import yaml
    cfg = yaml.load(open(os.path.join(ROOT, "runfiles", "YOND", src)).read(), Loader=yaml.FullLoader)
    cfg['arch']['nf'] = nf
    for k in ('dst', 'dst_eval', 'dst_test'):
        cfg[k]['root_dir'] = str(root_dir)
        cfg[k].update(dst_over)
    f = tmp_path / src
    f.write_text(yaml.dump(cfg))
    return str(f)


def P_stream_applies(drv):
    from yond_public_amd import pipeline as P
    return P.stream_applies(drv.pipe, drv.pipe, drv.biaslut)


def test_yond_any_full_frame_driver(tmp_path, monkeypatch):
    """N3: the `YOND_any`-style driver (README.md:38-47; runfiles/YOND/ANY_simple+full_pre_grumix.yml) on a directory of raw-DN
    `.npy` Bayer frames: black / white level and the ratio list from the runfile, whole-frame `iter` denoising, whole-frame
    PSNR / SSIM where a reference frame exists -- against the oracle's IterDenoise and block metrics on the same frames."""
    import yond_oracle as O
    from yond_public_amd import YOND_full as Y
    monkeypatch.chdir(tmp_path)
    frames = tmp_path / "frames"
    os.makedirs(frames / "gt")
    H, W, bl, wp = 192, 320, 63, 1023
    raws = []
    for k in range(2):
        noisy, clean = O.synth_noisy(H, W, 2.0, 12.0, 60 + k, clip=False)
        raw = np.round(noisy * (wp - bl) * 0.5 + bl).astype(np.float32)               # raw DN at half exposure (ratio 2 restores it)
        np.save(frames / f"f{k}.npy", raw)
        np.save(frames / "gt" / f"f{k}.npy", np.round(clean * (wp - bl) + bl).astype(np.float32))
        raws.append(raw)
    rf = _small_full_runfile(tmp_path, "ANY_simple+full_pre_grumix.yml", frames, H=H, W=W, ratio_list=[2])
    drv = Y.YOND_Full(['-f', rf, '-m', 'eval'])
    assert type(drv.dst_eval).__name__ == 'Any_Dataset' and len(drv.dst_eval) == 2
    assert Y.STREAM_EVAL and P_stream_applies(drv)                # (round 6: eval() feeds pipeline.denoise_stream -- per-frame parameter dicts, two lanes)
    res = drv.eval(-1)
    red = res['x2']
    assert red['count'] == 2
    streamed = {k: dict(v) for k, v in drv.metrics.items()}
    Y.STREAM_EVAL = False                                         # ... and one frame at a time: the same numbers
    try:
        res1 = drv.eval(-1)
    finally:
        Y.STREAM_EVAL = True
    assert abs(res1['x2']['psnr_last'] - red['psnr_last']) < 2e-3
    for k, m1 in drv.metrics.items():
        assert len(m1['reg']) == len(streamed[k]['reg'])
        np.testing.assert_allclose(np.asarray(m1['reg'][0], np.float64), np.asarray(streamed[k]['reg'][0], np.float64), rtol=1e-9)
        assert abs(m1['psnr'][-1] - streamed[k]['psnr'][-1]) < 2e-3
    drv.metrics = streamed
    arch = dict(drv.arch)
    sd = O.denoising_state_dict(arch, 0)
    torch.set_num_threads(8)
    pipe = dict(drv.pipe)
    ps = []
    for k in range(2):
        lr = ((raws[k] - bl) * 2 / (wp - bl)).astype(np.float32)
        p = dict(O.default_params(), wp=wp, bl=bl, ratio=2)
        p['scale'] = (wp - bl) / 2
        ref = O.IterDenoise(lr, arch, sd, pipe, p=p)
        m = drv.metrics[f'f{k}_x02']
        assert len(m['reg']) == len(ref['regs'])
        for it, (r, gr) in enumerate(zip(m['reg'], ref['regs'])):
            # (round 2 estimates from the two networks' outputs, which differ by ~1e-6: on a 192 x 320 frame the fit over the
            # few thousand selected pixels moves by up to 1e-4 relative)
            np.testing.assert_allclose(r[0], gr[0], rtol=2e-5 if it == 0 else 3e-4)
        hr = ((np.load(frames / "gt" / f"f{k}.npy") - bl) / (wp - bl)).astype(np.float32).clip(0, 1)
        want = O.psnr(np.asarray(ref['raw_dns'][-1], np.float32), hr)                  # whole-frame PSNR, data range 1
        assert abs(m['psnr'][-1] - want) < 2e-3
        ps.append(m['psnr'][-1])
    assert abs(red['psnr_last'] - np.mean(ps)) < 1e-9


def test_yond_eld_driver_on_a_converted_tree_and_synthetic_fallback(tmp_path, monkeypatch):
    """N3: the ELD runfile (cam_list x ratio_list) over a `.npy`-converted miniature tree; and, with no data at all, the
    synthetic stand-ins of the runfile's frame size."""
    from test_data_loader import _write_eld_tree
    from yond_public_amd import YOND_full as Y
    monkeypatch.chdir(tmp_path)
    root = tmp_path / "ELD"
    _write_eld_tree(root, cams=("SonyA7S2",), scenes=(1,), H=384, W=512)
    # (physically plausible contents instead of the loader test's random DN: long exposures = the clean scene, short ones =
    # Poisson-Gaussian frames at 1 / ratio of the exposure)
    import yond_oracle as O
    bl, wp = 512, 16383
    rng = np.random.default_rng(9)
    clean = O.synth_clean(384, 512).astype(np.float64) * 0.7
    for iso_id in range(3):
        for ratio_id, ratio in enumerate((1, 10, 100, 200)):
            lr_id = iso_id * 5 + ratio_id + 2
            e = clean * (wp - bl) / ratio
            K, sig = 8.0 * (iso_id + 1), 30.0                     # 14-bit DN
            raw = rng.poisson(e / K) * K + rng.normal(0, sig, e.shape) + bl
            np.save(root / "SonyA7S2" / "scene-1" / f"IMG_{lr_id:04d}.npy", raw.astype(np.float32))
    for hr_id in (1, 6, 11, 16):
        np.save(root / "SonyA7S2" / "scene-1" / f"IMG_{hr_id:04d}.npy", (clean * (wp - bl) + bl).astype(np.float32))
    rf = _small_full_runfile(tmp_path, "ELD_simple+full_pre_grumix.yml", root, cam_list=['SonyA7S2'], ratio_list=[1, 10], H=384, W=512)
    drv = Y.YOND_Full(['-f', rf, '-m', 'eval'])
    res = drv.eval(-1)
    assert set(res) == {'SonyA7S2 x1', 'SonyA7S2 x10'} and all(r['count'] == 3 for r in res.values())      # 1 scene x 3 ISOs per ratio
    assert all(np.isfinite(r['psnr_last']) for r in res.values())
    rf2 = _small_full_runfile(tmp_path, "LRID_simple+full_pre_grumix.yml", tmp_path / "nowhere", H=256, W=384, ratio_list=[1])
    drv2 = Y.YOND_Full(['-f', rf2, '-m', 'eval', '--synthetic', '2'])
    assert type(drv2.dst_eval).__name__ == 'SyntheticFrames'
    r2 = drv2.eval(-1)['x1']
    assert r2['count'] == 2 and r2['psnr_last'] > 20.0


def test_yond_sidd_benchmark_mode(tmp_path, monkeypatch):
    """`YOND_SIDD.py -m test` (YOND_SIDD.py:572-630, 742-744): the benchmark blocks through IterDenoise, the estimates kept as
    reg_test, the two submission arrays [N][32][256][256] written (first / last round)."""
    from yond_public_amd import YOND_SIDD as Y
    monkeypatch.chdir(tmp_path)
    Y.main(['-f', RUNFILE, '-m', 'test', '--synthetic', '2'])
    init = np.load(tmp_path / "npy" / "YOND_SIDD_simple+full_pre_grumix_iter" / "benchmark_init.npy")
    last = np.load(tmp_path / "npy" / "YOND_SIDD_simple+full_pre_grumix_iter" / "benchmark_results.npy")
    assert init.shape == last.shape == (2, 32, 256, 256)
    assert np.isfinite(last).all() and float(np.abs(last - init).max()) > 0          # round 2 ran and changed the result


def test_prefetcher_uploads_through_pinned_buffers_in_order():
    """data.Prefetcher on the GPU box: items read by loader threads, copied into reused pinned buffers and uploaded on the workers' own streams; the
    consumer's stream waits on each copy's event -- values, order and device placement are checked while the consumer keeps the GPU busy, with
    more items than pinned buffers per worker (each buffer is reused only after its previous upload has left it)."""
    import numpy as np
    from yond_public_amd.data import Prefetcher

    class Items:
        def __len__(self):
            return 24

        def __getitem__(self, k):
            rng = np.random.default_rng(k)
            return {'lr': rng.random((32, 64, 64), dtype=np.float32), 'hr': np.full((5, 7), k, np.uint16), 'lr_full': None, 'name': f'i{k}', 'k': k}
    busy = torch.zeros(1 << 24, device='cuda:0')
    seen = []
    for k, d in Prefetcher(Items(), list(range(24)), 'cuda:0', workers=3, depth=5):
        busy.add_(1.0)                                                     # (work on the consumer's stream between the items)
        assert d['lr'].is_cuda and d['lr'].dtype == torch.float32 and d['hr'].is_cuda and d['lr_full'] is None and d['k'] == k
        want = np.random.default_rng(k).random((32, 64, 64), dtype=np.float32)
        assert np.array_equal(d['lr'].cpu().numpy(), want) and float(d['hr'].max()) == k and float(d['hr'].min()) == k
        seen.append(k)
    assert seen == list(range(24)) and float(busy[0]) == 24.0
