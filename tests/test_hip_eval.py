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
