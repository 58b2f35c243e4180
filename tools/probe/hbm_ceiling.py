"""What plain streaming kernels reach on this box: fill (write only), copy (read + write), sum (read only) of a 388 MB tensor."""
import torch
n = 1504 * 2016 * 32
a = torch.empty(n, device='cuda'); b = torch.empty(n, device='cuda')


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


mb = n * 4 / 1e6
for name, fn, bytes_ in (("fill", lambda: a.fill_(1.0), mb), ("copy", lambda: b.copy_(a), 2 * mb), ("sum", lambda: a.sum(), mb),
                         ("add (2R+1W)", lambda: torch.add(a, b, out=b), 3 * mb)):
    s = t(fn)
    print(f"{name:12s} {s * 1e6:8.1f} us  {bytes_ / 1e6 / s:6.2f} TB/s")
