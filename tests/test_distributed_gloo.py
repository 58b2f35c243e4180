"""CPU, world_size 2, gloo: the image-parallel driver logic -- round-robin sharding, per-rank metric sums and
the single all-reduce at the end (the same code path runs over RCCL on the GPUs)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, os.environ["YOND_ROOT"])
    import torch
    from yond_public_amd import distributed as D
    rank, local, world = D.init(backend="gloo")
    assert world == 2
    mine = D.shard_indices(7, rank, world)
    sums = D.MetricSums(2)
    for k in mine:                                   # fake per-image metrics that depend only on k
        sums.update([40.0 + k, 41.0 + k], [0.9 + 0.01 * k, 0.91 + 0.01 * k])
    D.barrier()
    t = D.max_over_ranks(1.0 + rank)
    red = sums.reduce()
    if rank == 0:
        print("RESULT " + json.dumps({"red": red, "t": t, "mine": mine}))
''')


def test_two_rank_metric_reduction(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, YOND_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0]
    r = json.loads(line[7:])
    assert r["t"] == 2.0 and r["mine"] == [0, 2, 4, 6]
    ks = list(range(7))
    assert r["red"]["count"] == 7
    assert abs(r["red"]["psnr_iter0"] - sum(40.0 + k for k in ks) / 7) < 1e-12
    assert abs(r["red"]["psnr_last"] - sum(41.0 + k for k in ks) / 7) < 1e-12
    assert abs(r["red"]["ssim_iter1"] - sum(0.91 + 0.01 * k for k in ks) / 7) < 1e-12


def test_one_rank_torchrun_environment_still_creates_a_group(tmp_path):
    """WORLD_SIZE = 1 under torchrun: distributed.init() creates the process group anyway, and barrier / max-over-ranks /
    the metric reduction really issue collectives (counted in D.STATS) -- the same code an eight-rank RCCL job runs."""
    worker = textwrap.dedent('''
        import os, sys, json
        sys.path.insert(0, os.environ["YOND_ROOT"])
        from yond_public_amd import distributed as D
        import torch.distributed as dist
        rank, local, world = D.init(backend="gloo")
        assert world == 1 and dist.is_initialized() and D.launched_by_torchrun()
        sums = D.MetricSums(1)
        sums.update([40.0], [0.9])
        D.barrier()
        t = D.max_over_ranks(3.5)
        red = sums.reduce()
        print("RESULT " + json.dumps({"red": red, "t": t, "stats": D.STATS}))
        D.finalize()
        assert not dist.is_initialized()
    ''')
    script = tmp_path / "worker1.py"
    script.write_text(worker)
    env = dict(os.environ, YOND_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29532", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert r["t"] == 3.5 and r["red"]["count"] == 1 and r["red"]["psnr_last"] == 40.0
    assert (r["stats"]["all_reduce"], r["stats"]["barrier"], r["stats"]["backend"]) == (2, 1, "gloo")


def test_plain_process_has_no_group(monkeypatch):
    """Without the torchrun environment nothing is initialised and the helpers are local no-ops."""
    import sys as _sys
    _sys.path.insert(0, ROOT)
    from yond_public_amd import distributed as D
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    import torch.distributed as dist
    assert D.init() == (0, 0, 1) and not dist.is_initialized()
    assert D.max_over_ranks(2.0) == 2.0
    D.barrier()


EVAL_WORKER = textwrap.dedent('''
    import os, sys, json, types
    sys.path.insert(0, os.environ["YOND_ROOT"])
    import numpy as np
    import torch
    torch.cuda.synchronize = lambda *a, **k: None              # (no GPU here: only the DRIVER's logic runs -- the hot path is replaced below)
    from yond_public_amd import distributed as D, YOND_SIDD as Y, pipeline as P
    from yond_public_amd.data import SIDD_FRAME_HW
    rank, local, world = D.init(backend="gloo")

    class Items:
        """40 SIDD-validation-like items: 8 scenes from each of the five phones (five full-frame sizes), interleaved as the set is."""
        phones = list(SIDD_FRAME_HW)
        def __len__(self):
            return 40
        def item_size(self, k):
            h, w = SIDD_FRAME_HW[self.phones[k % 5]]
            return h * w
        def __getitem__(self, k):
            return {'lr': np.full((32, 2, 2), k, np.float32), 'hr': np.zeros((32, 2, 2), np.float32), 'lr_full': None, 'name': f'img_{k:03d}', 'cfa': 'rggb'}

    def fake_metrics(k, it):          # depends on the image only
        return 40.0 + 0.37 * k + it, 0.9 + 0.001 * k + 0.01 * it

    t = object.__new__(Y.YOND_SIDD)
    t.parser = types.SimpleNamespace(prefetch=4, loaders=2, group=int(os.environ.get("YOND_GROUP", "4")), verbose=False, stream=False)   # (the two-stream form needs a GPU)
    t.biaslut = None
    t.rank, t.local_rank, t.world = rank, local, world
    t.device = torch.device('cpu')
    t.pipe = {'iter': 'iter', 'max_iter': 1}
    t.logfile, t.method_name, t.dst_eval = None, 'stub', Items()

    def fake_one(data, params):
        k = int(data['lr'].reshape(-1)[0])
        rounds = 1 if k % 7 == 3 else 2                        # some images end at the beta1 < 0 guard (:445-447): -1 in iteration 1's meter
        return {'raw_dns': [torch.full((1,), float(10 * k + it)) for it in range(rounds)], 'regs': [(0.0, 0.0)] * rounds,
                'hr_raw': torch.zeros(1), 'lr_raw': None}
    t.IterDenoise = fake_one
    t.IterDenoiseGroup = lambda datas, plist: [fake_one(d, q) for d, q in zip(datas, plist)]

    def fake_block_metrics(dn, hr):
        v = int(dn[0].item())
        ps, ss = fake_metrics(v // 10, v % 10)
        return np.array([ps]), np.array([ss])
    P.block_metrics = fake_block_metrics
    mine = D.shard_dataset(t.dst_eval, rank, world)
    load = sum(t.dst_eval.item_size(k) for k in mine)
    loads = D.gather_over_ranks(float(load))
    counts = D.gather_over_ranks(float(len(mine)))
    red = t.eval(-1)
    if rank == 0:
        print("RESULT " + json.dumps({"red": red, "loads": loads, "counts": counts, "stats": {k: D.STATS[k] for k in ("all_reduce", "backend")}}))
    D.finalize()
''')


def test_eval_driver_on_eight_gloo_ranks(tmp_path):
    """YOND_SIDD.eval -- the REAL driver loop: size-aware sharding, loader threads, groups of images, per-image meters with the reference's
    -1 rule (:644-647), ONE metric all-reduce -- as EIGHT gloo ranks over 40 SIDD-like items of the five phones' frame sizes; only the hot
    path (IterDenoise / block metrics) is replaced by a deterministic stand-in.  The greedy shard's pixel loads lie within 10 % of each
    other, every image is evaluated exactly once, and the reduced means equal the single-process means to 1e-12."""
    import json
    script = tmp_path / "eval_worker.py"
    script.write_text(EVAL_WORKER)
    env = dict(os.environ, YOND_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29537", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert r["stats"]["backend"] == "gloo"
    assert sum(r["counts"]) == 40 and min(r["counts"]) >= 4
    assert max(r["loads"]) <= 1.10 * min(r["loads"]), r["loads"]
    ks = range(40)
    rounds = [1 if k % 7 == 3 else 2 for k in ks]
    f = lambda k, it: (40.0 + 0.37 * k + it, 0.9 + 0.001 * k + 0.01 * it)
    red = r["red"]
    assert red["count"] == 40
    assert abs(red["psnr_iter0"] - sum(f(k, 0)[0] for k in ks) / 40) < 1e-12
    assert abs(red["psnr_iter1"] - sum(f(k, 1)[0] if rounds[k] == 2 else -1.0 for k in ks) / 40) < 1e-12
    assert abs(red["ssim_iter1"] - sum(f(k, 1)[1] if rounds[k] == 2 else -1.0 for k in ks) / 40) < 1e-12
    assert abs(red["psnr_last"] - sum(f(k, rounds[k] - 1)[0] for k in ks) / 40) < 1e-12
    assert abs(red["ssim_last"] - sum(f(k, rounds[k] - 1)[1] for k in ks) / 40) < 1e-12
