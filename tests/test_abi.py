"""CPU: libyond_hip.so loads and exports every symbol include/yond_hip.h declares (no compute calls)."""
import os
import re
import ctypes

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "yond_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(yond_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "yond_pack_vst_norm_f32" in syms and "yond_conv2d_f32" in syms and len(syms) >= 25


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from yond_public_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    # the ctypes prototype table covers the same set
    assert sorted(_lib.PROTOTYPES) == declared_symbols()
    assert _lib.load().yond_abi_version() == 8


def test_host_side_argument_checks_without_gpu():
    from yond_public_amd import _lib
    lib = _lib.load()
    tn, kc = ctypes.c_int(), ctypes.c_int()
    assert lib.yond_conv_config(3, 1, 64, 64, 0, 0, 0, 0, ctypes.byref(tn), ctypes.byref(kc)) == 0 and (tn.value, kc.value) == (64, 16)
    assert lib.yond_conv_config(3, 2, 32, 64, 0, 0, 0, 0, ctypes.byref(tn), ctypes.byref(kc)) == 0 and (tn.value, kc.value) == (64, 8)
    assert lib.yond_conv_config(1, 1, 64, 128, 1, 0, 0, 0, ctypes.byref(tn), ctypes.byref(kc)) == 0 and tn.value == 32
    assert lib.yond_conv_config(5, 1, 64, 64, 0, 0, 0, 0, ctypes.byref(tn), ctypes.byref(kc)) == -2
    assert lib.yond_conv_config(3, 1, 24, 64, 0, 0, 0, 0, ctypes.byref(tn), ctypes.byref(kc)) == -2
    # deepest level of cfg 2 (94 x 126, 512 ch): 384 tiles of width 64 would leave half the last round idle
    assert lib.yond_conv_config(3, 1, 512, 512, 0, 1, 94, 126, ctypes.byref(tn), ctypes.byref(kc)) == 0 and tn.value == 32
    assert (tn.value, kc.value) == (32, 16)            # config A: one persistent workgroup per CU, 768 tiles = 3 rounds
    # level 3 (188 x 252, 256 ch): 1536 tiles of width 32 fill the 512 slots of config B (two workgroups per CU) exactly
    assert lib.yond_conv_config(3, 1, 256, 256, 0, 1, 188, 252, ctypes.byref(tn), ctypes.byref(kc)) == 0
    assert (tn.value, kc.value) == (32, 8)
    assert lib.yond_conv_config(3, 1, 32, 32, 0, 1, 1504, 2016, ctypes.byref(tn), ctypes.byref(kc)) == 0 and (tn.value, kc.value) == (32, 8)
    # null pointers are rejected before any launch
    assert lib.yond_pack_vst_norm_f32(None, 4, 4, None, 0, 0, 0, 0, 1, 1.0, 1.0, 0.0, 0.0, 1.0, None, None, 0, None, None) == -1
    assert lib.yond_conv2d_f32(None, None) == -1


def test_weight_packing_layout():
    """yond_pack_conv_weight_f32 produces [ct][chunk][tap][q][half][j][e] (the LDS image of conv.hip)."""
    import numpy as np
    from yond_public_amd import _lib
    lib = _lib.load()
    cout, cin, k, tn, kc = 64, 32, 3, 32, 16
    w = np.arange(cout * cin * k * k, dtype=np.float32).reshape(cout, cin, k, k)
    dst = np.empty(w.size, np.float32)
    assert lib.yond_pack_conv_weight_f32(w.ctypes.data, cout, cin, k, tn, kc, dst.ctypes.data) == 0
    d = dst.reshape(cout // tn, cin // kc, k * k, kc // 8, 2, tn, 4)
    for (ct, ch, tap, q, h, j, e) in [(0, 0, 0, 0, 0, 0, 0), (1, 1, 8, 1, 1, 31, 3), (1, 0, 4, 1, 0, 7, 2)]:
        co, ci = ct * tn + j, ch * kc + q * 8 + h * 4 + e
        assert d[ct, ch, tap, q, h, j, e] == w[co, ci, tap // 3, tap % 3]
    assert sorted(dst.tolist()) == sorted(w.reshape(-1).tolist())


def test_product_does_not_import_oracle():
    """The product package must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "yond_public_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "yond_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_cpu_tensor_is_rejected_loudly():
    import torch
    from yond_public_amd.archs import GuidedResUnet
    from yond_public_amd._lib import YondHipError
    net = GuidedResUnet(dict(name='GuidedResUnet', in_nc=4, out_nc=4, nf=8, nframes=1, res=True, norm=True))
    with pytest.raises(YondHipError):
        net(torch.zeros(1, 4, 32, 32), torch.tensor(0.1))
