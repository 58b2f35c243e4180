"""Run tests with every torch.empty / empty_like / new_empty float tensor POISONED (NaN-filled): a kernel that reads memory nobody wrote --
benign while the allocator hands back finite garbage, a NaN / range-guard trip when it does not -- then fails deterministically.
    python tools/probe/poison_run.py tests/test_hip_train.py -x -q [-k ...]"""
import sys
import torch

_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if t.is_floating_point() and t.device.type == 'cuda' and t.numel():
        t.fill_(float('nan'))
    return t


def empty(*a, **k):
    return _poison(_empty(*a, **k))


def empty_like(*a, **k):
    return _poison(_empty_like(*a, **k))


torch.empty, torch.empty_like = empty, empty_like
import pytest
sys.exit(pytest.main(sys.argv[1:]))
