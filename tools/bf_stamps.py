"""Diagnostic: cycle stamps of the fused estimator kernel (build with -DBF_STAMPS into a side library)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
side = os.path.join(ROOT, "gpurun_out", "libyond_dbg.so")
os.makedirs(os.path.dirname(side), exist_ok=True)
srcs = [os.path.join(ROOT, "yond_public_amd", "csrc", f) for f in sorted(os.listdir(os.path.join(ROOT, "yond_public_amd", "csrc"))) if f.endswith(".hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-DBF_STAMPS"] + sys.argv[1:] +
               ["-o", side] + srcs, check=True)
os.environ["YOND_HIP_LIB"] = side
import torch
from yond_public_amd import _lib as L
from yond_public_amd import pipeline as P
import yond_public_amd.synthetic as S
lib = L.load()
H, W = 3000, 4000
h, w = H // 2, W // 2
x = torch.from_numpy(S.synth_noisy(H, W, 4.0, 6.0, 0)[0]).cuda()
o = [torch.empty(4, h, w, device='cuda') for _ in range(3)]
q = np.ascontiguousarray(P.QUANTS)
ws = P._nle_workspace(4 * h * w, x.device)
for it in range(3):
    lib.yond_box_stats_self_fused_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), C.c_void_p(q.ctypes.data), len(q), L.ptr(ws), L.stream())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
lib.yond_box_stats_self_fused_f32(L.ptr(x), H, W, 29, 19, 0, L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), C.c_void_p(q.ctypes.data), len(q), L.ptr(ws), L.stream())
e1.record(); torch.cuda.synchronize()
print("kernel+memset: %.1f us" % (e0.elapsed_time(e1) * 1e3))
import ctypes
# dbg sits at the end of NleState
nst = 0
raw = ws.cpu().numpy()
# find dbg offset: state bytes = head..; we know sizeof via yond_nle_ws_bytes(0) - pads
tot = int(lib.yond_nle_ws_bytes(0)) - 16 * 64 - 256
tot = (tot // 256) * 256
for guess in range(tot - 256 - 512, tot + 8, 8):
    pass
dbg = None
# dbg is the last 512 bytes of the struct (before padding to 256): scan backwards for the non-zero block
arr = raw[:tot].view(np.uint64)
nz = np.nonzero(arr[-200:])[0]
blk = arr[-200:][nz.min():nz.min() + 64] if len(nz) else None
print("stamps per wave [stage1, stage2, stats+bins, stores, bar1, task, bar2] (cycles, whole kernel):")
if blk is not None:
    for wv in range(8):
        print(wv, [int(v) for v in blk[wv * 8:wv * 8 + 7]])
