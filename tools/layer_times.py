"""Per-launch times of one SNR-Net forward (HIP events around every convolution launch): at the cfg-2 shape by default,
`python tools/layer_times.py B H W` for a batch of B packed [H][W] inputs (cfg 3: 32 128 128)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yond_public_amd import archs as A, synthetic as S, pipeline as P
from yond_public_amd import engine as E
if os.environ.get("K1_D2_LEVELS") is not None:
    E.K1_D2_LEVELS = tuple(int(v) for v in os.environ["K1_D2_LEVELS"].split(",") if v)
if os.environ.get("K1_D2_LEVELS_HALF") is not None:
    E.K1_D2_LEVELS_HALF = tuple(int(v) for v in os.environ["K1_D2_LEVELS_HALF"].split(",") if v)
if os.environ.get("SP_CONV1_MIN_LEVEL"):            # A/B of the engine's data-flow constant (tools only)
    E.SP_CONV1_MIN_LEVEL = int(os.environ["SP_CONV1_MIN_LEVEL"])
arch = dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True)
net = A.GuidedResUnet(dict(arch)); net.load_state_dict(S.procedural_state_dict(net, 0)); net = net.to('cuda').eval()
if os.environ.get("PRECISION"):                    # 'fp16': the BASELINE cfg 5 path (python tools/layer_times.py 1 2016 3008)
    net.precision = os.environ["PRECISION"]
plan = P._plan_of(net, torch.device('cuda'))
B, Hh, Ww = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 1504, 2016)
x = torch.rand(B, Hh, Ww, 4, device='cuda'); t = torch.full((B,), 0.03, device='cuda'); ub = x.reshape(B, -1).max(1).values.contiguous()
for _ in range(3): plan.forward_nhwc4(x, t, ub=ub)
torch.cuda.synchronize()
acc = {}
for rep in range(5):
    plan.prof = []
    plan.forward_nhwc4(x, t, ub=ub)
    torch.cuda.synchronize()
    for i, (tag, fl, e0, e1) in enumerate(plan.prof):
        acc.setdefault((i, tag), []).append(e0.elapsed_time(e1) * 1e3)
    plan.prof = None
tot = 0
for (i, tag), v in sorted(acc.items()):
    m = sorted(v)[len(v) // 2]
    tot += m
    print(f"{i:2d} {tag:42s} {m:8.1f} us")
print("sum of conv launches: %.1f us" % tot)
