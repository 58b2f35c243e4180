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
