"""Phase timestamps of the fused level-0 block kernel (csrc/block0_fused.hip, debug build -DB0_DBG=1 -> tools/probe/libyond_b0dbg.so):
cycles per phase of workgroup 0, waves 0 and 4, over its first tiles.
    python tools/b0_dbg.py build          (CPU box)
    python tools/b0_dbg.py [H W]          (GPU box; default 1504 2016)"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
DBG = os.path.join(HERE, "probe", "libyond_b0dbg.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from yond_public_amd.build import build_lib
    print(build_lib(extra_flags=["-DB0_DBG=1"], lib=DBG))
    sys.exit(0)
import numpy as np, torch
os.environ["YOND_HIP_LIB"] = DBG
from yond_public_amd import _lib as L
from yond_public_amd.engine import _PackedConv, DenoiserPlan
plan = DenoiserPlan.__new__(DenoiserPlan); plan.lib, plan.dev, plan.prof = L.load(), torch.device('cuda:0'), None
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1504, 2016)
g = torch.Generator().manual_seed(0)
pc1 = _PackedConv(plan.dev, torch.randn(32, 32, 3, 3, generator=g) / 17, None, 3, 1, [32])
pc2 = _PackedConv(plan.dev, torch.randn(32, 32, 3, 3, generator=g) / 17, None, 3, 1, [32])
f = [torch.randn(1, 32, device='cuda') for _ in range(4)]
x = torch.randn(1, 8, H * W, 4, device='cuda')
out = plan._new_sp('o', 1, H, W, 32)
dll = C.CDLL(DBG)
rd = dll.yond_block0_debug_read; rd.argtypes = [C.c_void_p]; rd.restype = C.c_int
for _ in range(3):
    plan._block0(pc1, pc2, x, 1, H, W, f, 2, dst=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    plan._block0(pc1, pc2, x, 1, H, W, f, 2, dst=out)
e1.record(); torch.cuda.synchronize()
print("launch: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 5))
buf = np.zeros((2, 32, 12), np.uint64)
assert rd(buf.ctypes.data) == 0
t = buf.astype(np.int64)
names = ["wait-in", "conv1", "bar", "epi1", "issue", "bar", "conv2", "epi2", "bar", "stage", "loop"]
for wv in range(2):
    print(" wave", wv * 4, ": tile | " + " ".join(f"{n:>7s}" for n in names) + " |  total")
    for ti in range(2, 26):
        r = t[wv, ti]; nxt = t[wv, ti + 1][0]
        d = [r[i + 1] - r[i] for i in range(10)] + [nxt - r[10]]
        print("      %3d | " % ti + " ".join("%7d" % v for v in d) + " | %6d" % (nxt - r[0]))
