"""Phase timestamps of the split-plane kernels (debug build tools/probe/libyond_sdbg.so, -DSPLIT_DBG=1): cycles per phase of
workgroup 0, waves 0 and 4, for conv1 (register-staged input, split-plane store: reader `osp`) and conv2 (LDS-DMA input: `isp`).
    python tools/split_dbg_sp.py build            (on the CPU box: compiles the debug library)
    python tools/split_dbg_sp.py C H W            (on the GPU box)"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
DBG = os.environ.get("SDBG_LIB", os.path.join(HERE, "probe", "libyond_sdbg.so"))      # (SDBG_LIB: another debug build, e.g. -DYOND_SPLIT_ASYM=0)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from yond_public_amd.build import build_lib
    print(build_lib(extra_flags=["-DSPLIT_DBG=1"], lib=DBG))
    sys.exit(0)
import numpy as np, torch
os.environ["YOND_HIP_LIB"] = DBG
from yond_public_amd import _lib as L
from yond_public_amd.engine import _PackedConv, DenoiserPlan
lib = L.load()
plan = DenoiserPlan.__new__(DenoiserPlan); plan.lib, plan.dev, plan.prof = lib, torch.device('cuda:0'), None
Cc, h, w = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 376, 504)
g = torch.Generator().manual_seed(0)
pc = _PackedConv(plan.dev, torch.randn(Cc, Cc, 3, 3, generator=g) / (3 * Cc ** 0.5), torch.randn(Cc, generator=g), 3, 1, [Cc])
x = torch.randn(1, h, w, Cc, device='cuda'); dst = torch.empty(1, h, w, Cc, device='cuda')
es = torch.randn(1, Cc, device='cuda'); et = torch.randn(1, Cc, device='cuda')
tsp = plan._new_sp('t', 1, h, w, Cc)
dll = C.CDLL(DBG)


def show(reader, title):
    f = getattr(dll, reader); f.argtypes = [C.c_void_p]; f.restype = C.c_int
    buf = np.zeros((2, 64, 12), np.uint64)
    assert f(buf.ctypes.data) == 0
    t = buf.astype(np.int64)
    print(title)
    for wv in range(2):
        print(" wave", wv * 4, ": step | head  mfma-stretch (quarters)  post(staging/epilogue)  barrier  tail | total")
        tot = []
        for sidx in range(2, 26):
            r = t[wv, sidx]; nxt = t[wv, sidx + 1][0]
            epi = "   epilogue %d + barrier %d" % (r[6] - r[4], r[7] - r[6]) if r[6] > r[4] and r[5] - r[4] > 600 else ""
            tot.append(nxt - r[0])
            print("   %3d | %5d  %6d (%5d %5d %5d %5d)  %5d  %5d  %5d | %6d%s" % (sidx, r[1] - r[0], r[2] - r[1], r[8] - r[1], r[9] - r[8], r[10] - r[9],
                  r[2] - r[10], r[3] - r[2], r[4] - r[3], r[5] - r[4], nxt - r[0], epi))
        print("   mean cycles per step: %.0f" % (sum(tot) / len(tot)))


for _ in range(3):
    plan._conv(pc, x, None, 1, h, w, tsp, escale=es, eshift=et, ebatch=1, pre_act=1, post_act=1, algo='split', out_fmt=1)
torch.cuda.synchronize()
show("yond_split_debug_read_wres" if Cc == 32 else "yond_split_debug_read_osp", "conv1: register-staged input, split-plane store")
for _ in range(3):
    plan._conv(pc, tsp, None, 1, h, w, dst, escale=es, eshift=et, ebatch=1, res=x, algo='split', in_fmt=1)
torch.cuda.synchronize()
show("yond_split_debug_read_isp", "conv2: split-plane input by LDS-DMA, residual, NHWC store")
out = plan._new_sp('o', 1, h, w, Cc)
x4p = x.reshape(1, h * w, Cc // 4, 4).permute(0, 2, 1, 3).contiguous()
for _ in range(3):
    plan._conv(pc, tsp, None, 1, h, w, out, escale=es, eshift=et, ebatch=1, res=x4p, algo='split', in_fmt=1, out_fmt=1, res_fmt=2)
torch.cuda.synchronize()
show("yond_split_debug_read_wres" if Cc == 32 else "yond_split_debug_read_isp_osp", "conv2 of the flow: split-plane input, planes-of-4 residual, split-plane store")
# the stride-2 layer of the same level in the flow: split planes in -> planes of 4 channels out
pc2 = _PackedConv(plan.dev, torch.randn(2 * Cc, Cc, 3, 3, generator=g) / (3 * Cc ** 0.5), torch.randn(2 * Cc, generator=g), 3, 2, [Cc])
nxt = torch.empty(1, h // 2, w // 2, 2 * Cc, device='cuda')
for _ in range(3):
    plan._conv(pc2, out, None, 1, h, w, nxt, algo='split', in_fmt=1, out_fmt=2)
torch.cuda.synchronize()
show("yond_split_debug_read_isp_k1s2", "stride 2: split-plane input through the register pipeline, planes-of-4 store")
# the decoder GEMM of the level below (input c 2 channels at (h/2, w/2) + skip tensor c channels at (h, w) -> c channels at (h, w)): short steps
# (three 16-channel pseudo-taps = 18 MFMAs per wave), register-staged split planes
if Cc >= 64:
    hl, wl = h // 2, w // 2
    cur = plan._new_sp('cur', 1, hl, wl, 2 * Cc)
    skip = plan._new_sp('skip', 1, 2 * hl, 2 * wl, Cc)
    pcu = _PackedConv(plan.dev, torch.randn(3 * Cc, Cc, 2, 2, generator=g) / (3 * Cc) ** 0.5, torch.randn(Cc, generator=g), 1, 1, [2 * Cc, Cc], shuffle=True)
    up = torch.empty(Cc * 4 * hl * wl, device='cuda')
    for _ in range(3):
        plan._conv(pcu, cur, skip, 1, hl, wl, up, algo='split', in_fmt=1, out_fmt=2)
    torch.cuda.synchronize()
    show("yond_split_debug_read_isp_k1s2", "decoder GEMM: split-plane inputs through the register pipeline, planes-of-4 pixel-shuffle store")
