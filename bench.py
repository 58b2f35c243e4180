#!/usr/bin/env python
"""bench.py -- YOND hot path on MI355X: Bayer megapixels/s end-to-end (NLE + VST + denoise + iVST).

    python bench.py [--gpus N --steps K --warmup W]            (N > 1: launched by torch.distributed.run)

A step is one pass of the whole per-image path over one synthetic 3000 x 4000 Bayer frame that is already
resident in HBM (BASELINE.json configs[1]): self-calibrated noise-level estimation, bias-LUT build,
pack+VST, SNR-Net (GuidedResUnet nf=32; fp32 results, convolutions as fp32-accurate split-operand products on the fp16
MFMA -- `--precision fp32-mfma` keeps them on the fp32-input MFMA) forward, inverse VST+unpack.  For the 'once' pipeline
the K steps run through pipeline.denoise_stream: the NLE of frame k+1 on a second HIP stream under the convolutions
of frame k (`--sequential`: one frame at a time).  Frames are sharded one
per GPU (image parallel, weak scaling); there is no data-path collective, the only RCCL traffic is the
barrier / max-over-ranks of the timing and the final PSNR reduction.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from yond_public_amd import distributed as D          # noqa: E402
from yond_public_amd import pipeline as P             # noqa: E402
from yond_public_amd import synthetic as S            # noqa: E402
from yond_public_amd import archs as A                # noqa: E402
from yond_public_amd import _lib as L                 # noqa: E402

PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
PEAK_HBM_GBPS = 8000.0

ARCHS = {
    'GuidedResUnet': dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
    'UNetSeeInDark': dict(name='UNetSeeInDark', in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
}


def cpu_baseline(arch, mode, seed, threads):
    """The oracle (CPU restatement of the reference path) timed on the host, on a bounded sample of the
    same workload.  Test infrastructure used ONLY as the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import yond_oracle as O
    H, W = 2048, 3072
    noisy, _ = O.synth_noisy(H, W, 4.0, 6.0, 0)
    sd = O.procedural_state_dict(arch, seed)
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': mode, 'max_iter': 1, 'full_dn': True,
            'collab_sidd256': False}
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    O.IterDenoise(noisy, arch, sd, pipe)
    dt = time.perf_counter() - t0
    torch.set_num_threads(old)
    return {"value": H * W / 1e6 / dt, "unit": "Bayer MP/s", "cores": threads, "kind": "port",
            "sample": f"one {H}x{W} synthetic Bayer frame, same pipeline ('{mode}'), oracle/yond_oracle.py "
                      f"(NumPy/SciPy + PyTorch-CPU), {dt:.1f} s"}


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE collected separately, gfx950 corrections applied by tools/pmc_traffic.py); None if not collected."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        t = json.load(open(path)).get(kernel)
        return t and {"hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "unit": "B", "source": "profiles/r01_pmc_traffic.json"}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # 20 frames of 5 ms: the first frame of a stream has no overlap partner
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="once", choices=["once", "iter"])
    ap.add_argument("--arch", default="GuidedResUnet", choices=list(ARCHS))
    ap.add_argument("--height", type=int, default=3000)
    ap.add_argument("--width", type=int, default=4000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sequential", action="store_true", help="one frame at a time (IterDenoise) instead of the two-stream driver")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiments: no HIP events around the kernels (no roofline objects)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32-mfma", "fp16"],
                    help="fp32 (headline): fp32 results, 3x3 convolutions as fp32-accurate split-operand products on the fp16 MFMA; "
                         "fp32-mfma: every convolution on the fp32-input MFMA; fp16: BASELINE cfg 5 (not the headline configuration)")
    a = ap.parse_args()

    rank, local, world = D.init()
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    L.load()

    arch = ARCHS[a.arch]
    net = getattr(A, arch['name'])(dict(arch, precision=a.precision))
    net.load_state_dict(S.procedural_state_dict(net, 0))
    net = net.to(dev).eval()
    H, W = a.height, a.width
    noisy, clean = S.synth_noisy(H, W, 4.0, 6.0, rank)
    frame = torch.from_numpy(noisy).to(dev)
    clean_d = torch.from_numpy(clean).to(dev)
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': a.mode, 'max_iter': 1, 'full_dn': True,
            'collab_sidd256': False}

    def step():
        return P.IterDenoise(frame, net, arch, pipe)

    def run(nsteps):
        """nsteps passes of the hot path; 'once' mode through the two-stream driver (the NLE of frame k+1 overlaps the
        convolutions of frame k), unless --sequential."""
        last = None
        if a.sequential:
            for _ in range(nsteps):
                last = step()
        else:
            for last in P.denoise_stream((frame for _ in range(nsteps)), net, arch, pipe):
                pass
        return last

    res = run(a.warmup)
    plan = P._plan_of(net, dev)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    # HIP events (on the launch stream) bracket every launch of the 3x3 stride-1 convolution kernels -- the dominant
    # family: 18 launches per forward, 91 % of the MACs -- inside the timed region; an event pair costs the stream a few
    # microseconds, so the other kernels and the VST / NLE stages are timed in one more, instrumented pass behind it
    is33 = lambda t: t.startswith("conv_mfma_kernel<3,1") or t.startswith("conv_wino_kernel") or t.startswith("conv_split_kernel<1,")
    plan.prof = None if a.no_kernel_events else []
    plan.prof_only = is33
    t0 = time.perf_counter()
    res = run(a.steps)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    prof, plan.prof = plan.prof or [], None
    plan.prof_only = None
    stage_prof = []
    prof_all = []
    if not a.no_kernel_events:
        plan.prof = []
        P.PROF = []
        step()
        torch.cuda.synchronize()
        prof_all, plan.prof = plan.prof, None
        stage_prof, P.PROF = P.PROF, None
    stage_steps = 1

    # dominant kernel: the 3x3 stride-1 convolution kernel that takes the most time (18 launches per forward, 91 % of the
    # MACs: the split-operand kernel's 64-channel-tile shape; with --precision fp32-mfma the Winograd kernel)
    per = {}
    for tag, flops, e0, e1 in prof:
        ms = e0.elapsed_time(e1)
        k = per.setdefault(tag, [0, 0.0, 0.0])
        k[0] += 1
        k[1] += ms
        k[2] += flops
    dom = max((t for t in per if is33(t)), key=lambda t: per[t][1], default=None)
    PEAK = PEAK_F16_MFMA_TFLOPS if (a.precision == "fp16" or (dom or "").startswith("conv_split_kernel")) else PEAK_F32_MFMA_TFLOPS
    roof = None
    if dom:
        n, ms, fl = per[dom]
        ach = fl / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK, "unit": "TFLOP/s",
                "frac": round(ach / PEAK, 4), "traffic": pmc_traffic(dom), "launches": n,
                "avg_launch_ms": round(ms / n, 4), "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3)}
        if dom.startswith("conv_wino_kernel"):
            # Winograd F(2x2,3x3) issues 16 multiplications per patch where the direct algorithm has 36: `achieved`
            # counts the ALGORITHMIC flops of the convolution (it can exceed the matrix-core peak); the flops the MFMA
            # unit really executes are 16/36 of that
            roof["algorithm"] = "winograd F(2x2,3x3): 16/36 of the direct multiplications, fp32 throughout"
            roof["mfma_issued_tflops"] = round(ach * 16.0 / 36.0, 2)
            roof["mfma_issued_frac"] = round(ach * 16.0 / 36.0 / PEAK_F32_MFMA_TFLOPS, 4)
        if dom.startswith("conv_split_kernel") and dom.endswith(",2>"):
            # every fp32 product a*w is evaluated as h_a h_w + 2^-11 (h_a l_w + l_a h_w) by THREE fp16 MFMAs (fp32
            # accumulate): `achieved` counts the ALGORITHMIC flops of the convolution, the matrix cores execute 3x that
            roof["algorithm"] = ("direct 3x3, fp32 operands split into two fp16 halves when staged into LDS (22-23 significant "
                                 "bits), 3 v_mfma_f32_32x32x16_f16 per fp32 product block, fp32 accumulate")
            roof["mfma_issued_tflops"] = round(3.0 * ach, 2)
            roof["mfma_issued_frac"] = round(3.0 * ach / PEAK_F16_MFMA_TFLOPS, 4)
            roof["achieved_over_fp32_mfma_peak"] = round(ach / PEAK_F32_MFMA_TFLOPS, 4)
        others = {t: {"launches": v[0], "avg_launch_ms": round(v[1] / v[0], 4), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2)}
                  for t, v in per.items() if t != dom and is33(t)}
        if others:
            roof["other_3x3_kernels"] = others
    # VST + NLE stages against the HBM roofline: algorithmic bytes of SURVEY section 8(d) (24 B per Bayer pixel for a
    # 'once' pass: K1 8 + K4 8 + self-NLE 8; 'iter' adds a second K1/K4 pass and the collab NLE: 56 B) over the summed
    # stage times (HIP events on the launch stream; the host gaps between the kernels of a stage are inside)
    stage_ms = {}
    for tag, e0, e1 in stage_prof:
        stage_ms[tag] = stage_ms.get(tag, 0.0) + e0.elapsed_time(e1)
    roof_hbm = None
    if stage_ms:
        tot_ms = sum(stage_ms.values()) / stage_steps
        bpp = 24.0 if a.mode == "once" else 56.0
        gbs = bpp * H * W / (tot_ms * 1e-3) / 1e9
        roof_hbm = {"stage": "VST+NLE (K1, K4, K5-K7)", "bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(gbs / 8000.0, 4), "algorithmic_bytes_per_bayer_px": bpp, "ms_per_step": round(tot_ms, 4),
                    "stage_ms_per_step": {k: round(v / stage_steps, 4) for k, v in stage_ms.items()},
                    "measured": "HIP events, one instrumented pass after the timed region"}
    conv_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in prof_all)
    conv_fl = sum(fl for _, fl, _, _ in prof_all)

    # the same job with every convolution on the fp32-input MFMA (Winograd / direct kernels), for reference beside the headline
    strict = None
    if a.precision == "fp32":
        net2 = getattr(A, arch['name'])(dict(arch, precision='fp32-mfma'))
        net2.load_state_dict(S.procedural_state_dict(net2, 0))
        net2 = net2.to(dev).eval()
        r2 = P.IterDenoise(frame, net2, arch, pipe)
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        for _ in range(2):
            r2 = P.IterDenoise(frame, net2, arch, pipe)
        torch.cuda.synchronize()
        D.barrier()
        el2 = D.max_over_ranks(time.perf_counter() - t1, dev)
        dmax = (r2['raw_dns'][-1] - res['raw_dns'][-1]).abs().max().item()
        strict = {"value": round(world * 2 * H * W / 1e6 / el2, 2), "unit": "Bayer MP/s", "ms_per_step": round(el2 / 2 * 1e3, 3),
                  "steps": 2, "max_abs_output_difference_to_headline_path": dmax}
        del net2, r2

    # final metric reduction (the only collective of the eval path): PSNR of the last output vs the clean frame
    dn = res['raw_dns'][-1]
    mse = torch.mean((dn.double() - clean_d.double()) ** 2).item()
    sums = D.MetricSums(1)
    sums.update([10 * np.log10(1.0 / mse)], [0.0])
    red = sums.reduce(dev)

    if rank == 0:
        mp = H * W / 1e6
        out = {
            "metric": "Bayer megapixels/sec end-to-end (NLE+VST+denoise+iVST)",
            "value": round(world * a.steps * mp / elapsed, 2), "unit": "Bayer MP/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp32": "f32 (3x3 convolutions: f32 operands as two f16 halves on the f16 MFMA, 3 products per f32 product, "
                              "f32 accumulate; error vs float64 <= the f32-MFMA kernels')",
                      "fp32-mfma": "f32", "fp16": "f16 MFMA operands, f32 accumulate and tensors (cfg 5)"}[a.precision],
            "data": "synthetic",
            "config": {"workload": f"configs[{4 if a.precision == 'fp16' else 1}]: one {H}x{W} synthetic Poisson-Gaussian Bayer frame per GPU, full "
                                   f"NLE+VST+{a.arch}(nf=32)+iVST, pipeline '{a.mode}', bias_corr=pre, k=29",
                       "frames_per_step_per_gpu": 1, "parallelism": f"image-parallel x{world}",
                       "driver": "IterDenoise, one frame at a time" if (a.sequential or a.mode != "once") else
                                 "denoise_stream: NLE of frame k+1 on a second HIP stream while the convolutions of frame k run"},
            "roofline": roof,
            "fp32_mfma_path": strict,
            "roofline_vst_nle": roof_hbm,
            "conv_stack": {"ms_per_step": round(conv_ms, 3), "tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None,
                           "share_of_step": round(conv_ms / (elapsed / a.steps * 1e3), 3) if conv_ms else None,
                           "measured": "HIP events around every convolution launch, one instrumented pass after the timed region"},
            "psnr_vs_clean_db": round(red["psnr_last"], 3),
            "estimated_K_sigma": [round(float(v), 4) for v in res['params'][-1]],
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(arch, a.mode, 0, 1)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
