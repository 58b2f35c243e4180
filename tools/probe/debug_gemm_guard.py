"""One-off: which yond_gemm_split_f32 call trips the x-range guard in the trainer test (run on the GPU box).  Every call gets its own pair
of status words (OR-ed into the plan's real words on the device, so the step's behaviour is unchanged); after a failure the words tell
which calls tripped."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from yond_public_amd import train as T
orig = T._gemm_split
BUF = torch.zeros(4096, dtype=torch.int32, device='cuda')
calls = []
def wrapped(plan, srcs, P, n_p, n_real, sn_lo, sn_hi, nblk, bias, y, ldy, shuffle=0, H=0, W=0):
    i = len(calls) % 1000
    real = plan.status
    mine = BUF[4 * i:4 * i + 4]
    mine.zero_()
    plan.status = mine
    try:
        out = orig(plan, srcs, P, n_p, n_real, sn_lo, sn_hi, nblk, bias, y, ldy, shuffle, H, W)
    finally:
        plan.status = real
    real[0:2].bitwise_or_(mine[0:2])
    calls.append((i, P, n_p, n_real, shuffle, [(tuple(s[0].shape), s[4], s[5], s[7], s[0]) for s in srcs], y))
    return out
T._gemm_split = wrapped
import test_hip_train as TT
from yond_public_amd import trainer_AWGN as TA
import pathlib
for attempt in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    tmp = pathlib.Path(tempfile.mkdtemp()); os.chdir(tmp)
    torch.manual_seed(11)
    calls.clear()
    rf, cfg = TT._small_train_runfile(tmp, "GRU_5to50_norm_mix.yml")
    try:
        TA.main(['-f', rf, '-m', 'train', '--synthetic', '16'])
        print(f"run {attempt}: trainer finished without a fatal trip", flush=True)
    except Exception as e:
        print(f"run {attempt}: FAILED:", type(e).__name__, str(e)[:200], flush=True)
        st = BUF.cpu().reshape(-1, 4)
        for (i, P, n_p, n_real, shuffle, srcs, y) in calls[-60:]:
            if int(st[i, 0]) & 1 or int(st[i, 1]) & 1:
                desc = []
                for (shape, ld, k, k_real, x) in srcs:
                    xv = x.reshape(-1)[:P * ld].reshape(P, ld)[:, :k]
                    desc.append(f"x{shape} ld={ld} k={k} k_real={k_real} absmax={float(xv.abs().max()):.4g} nan={bool(torch.isnan(xv).any())} "
                                f"pad-absmax={float(xv[:, k_real:].abs().max()) if k_real < k else 0:.4g}")
                print(f"  [trip] call slot {i}: words {st[i, :2].tolist()} P={P} n_p={n_p} n_real={n_real} shuffle={shuffle} " + " | ".join(desc), flush=True)
        break
