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
