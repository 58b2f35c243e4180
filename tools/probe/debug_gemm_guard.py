"""One-off: which yond_gemm_split_f32 call trips the new x-range guard in the trainer test (run on the GPU box)."""
import os, sys, tempfile, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from yond_public_amd import train as T
orig = T._gemm_split
seen = [0]
def wrapped(plan, srcs, P, n_p, n_real, sn_lo, sn_hi, nblk, bias, y, ldy, shuffle=0, H=0, W=0):
    if torch.cuda.is_current_stream_capturing():
        return orig(plan, srcs, P, n_p, n_real, sn_lo, sn_hi, nblk, bias, y, ldy, shuffle, H, W)
    before = int(plan.status[0])
    out = orig(plan, srcs, P, n_p, n_real, sn_lo, sn_hi, nblk, bias, y, ldy, shuffle, H, W)
    after = int(plan.status[0])
    if after & 1 and not before & 1 and seen[0] < 6:
        seen[0] += 1
        for (x, wp, sk_lo, sk_hi, ld, k, kblk, k_real) in srcs:
            xv = x.reshape(-1)[:P * ld].reshape(P, ld)[:, :k]
            print(f"[trip] P={P} n_p={n_p} n_real={n_real} shuffle={shuffle} ld={ld} k={k} k_real={k_real} x.shape={tuple(x.shape)} contiguous={x.is_contiguous()} "
                  f"absmax={float(xv.abs().max()):.4g} nan={bool(torch.isnan(xv).any())} real-absmax={float(xv[:, :k_real].abs().max()):.4g} numel={x.numel()} P*ld={P * ld} y-nan={bool(torch.isnan(y).any())} y-absmax={float(y.abs().max()):.4g}", flush=True)
    return out
T._gemm_split = wrapped
import test_hip_train as TT
from yond_public_amd import trainer_AWGN as TA
import pathlib
tmp = pathlib.Path(tempfile.mkdtemp()); os.chdir(tmp)
torch.manual_seed(11)
rf, cfg = TT._small_train_runfile(tmp, "GRU_5to50_norm_mix.yml")
try:
    TA.main(['-f', rf, '-m', 'train', '--synthetic', '16'])
    print("trainer finished without a fatal trip")
except Exception as e:
    print("FAILED:", type(e).__name__, str(e)[:300])
