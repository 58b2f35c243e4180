"""Phase timestamps of the split-operand kernel (debug build tools/probe/libyond_sdbg.so): cycles per phase, wave 0 / wave 4 of workgroup 0."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YOND_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libyond_sdbg.so")
os.environ["YOND_CONV_WINO"] = "split"
from yond_public_amd import _lib as L
from yond_public_amd.engine import _PackedConv, DenoiserPlan
lib = L.load()
plan = DenoiserPlan.__new__(DenoiserPlan); plan.lib, plan.dev, plan.prof = lib, torch.device('cuda:0'), None
Cc, h, w = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 376, 504)
g = torch.Generator().manual_seed(0)
pc = _PackedConv(plan.dev, torch.randn(Cc, Cc, 3, 3, generator=g) / (3 * Cc ** 0.5), torch.randn(Cc, generator=g), 3, 1, [Cc])
x = torch.randn(1, h, w, Cc, device='cuda'); dst = torch.empty(1, h, w, Cc, device='cuda')
for _ in range(3):
    plan._conv(pc, x, None, 1, h, w, dst, pre_act=1)
torch.cuda.synchronize()
buf = np.zeros((2, 64, 12), np.uint64)
lib._FuncPtr  # noqa
f = C.CDLL(os.environ["YOND_HIP_LIB"]).yond_split_debug_read
f.argtypes = [C.c_void_p]; f.restype = C.c_int
print("rc", f(buf.ctypes.data))
t = buf.astype(np.int64)
for wv in range(2):
    print("wave", wv * 4)
    print(" step issue mfma+stage  -   bar   epi  | step total")
    for sidx in range(2, 40):
        r = t[wv, sidx]
        nxt = t[wv, sidx + 1][0]
        extra = "  epilogue %d, barrier behind it %d, rest %d" % (r[6] - r[4], r[7] - r[6], r[5] - r[7]) if r[6] > r[4] and r[5] - r[4] > 1000 else ""
        q = "  quarters of the MFMA stretch %d %d %d %d" % (r[8] - r[1], r[9] - r[8], r[10] - r[9], r[2] - r[10])
        print(" %3d %6d %6d %6d %5d %5d | %6d%s%s" % (sidx, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], nxt - r[0], q, extra))
