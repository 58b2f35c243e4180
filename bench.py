#!/usr/bin/env python
"""bench.py -- YOND hot path on MI355X: Bayer megapixels/s end-to-end (NLE + VST + denoise + iVST).

    python bench.py [--gpus N --steps K --warmup W]

`--gpus N` with N > 1 and no torchrun environment: this process starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD (before anything touches the GPU) and
relays rank 0's JSON line; under torchrun (RANK / WORLD_SIZE set) it is one rank of the job.

A step is one pass of the whole per-image path over one BATCH of synthetic 3000 x 4000 Bayer frames that are already
resident in HBM (BASELINE.json configs[1]; `--frames-per-step`, default 24, so that the K = 20 timed steps of the driver
span >= 2 s): per frame the self-calibrated noise-level estimation, bias-LUT build, pack+VST, SNR-Net forward
(GuidedResUnet nf=32; fp32 results, convolutions as fp32-accurate split-operand products on the fp16 MFMA --
`--precision fp32-mfma` keeps them on the fp32-input MFMA), inverse VST + unpack.  'once' pipelines stream the frames
through pipeline.denoise_stream (NLE of frame k+1 on a second HIP stream under the convolutions of frame k); the
per-frame latency of SURVEY section 8d (one frame at a time, resident -> resident) is measured right behind the timed region
and reported as `sequential` on the same line.  `--mode iter`: the shipped two-round pipeline (YOND_SIDD.py:419-472), one
frame at a time, asserted to run BOTH rounds.  Frames are sharded one stream per GPU (image parallel, weak scaling);
there is no data-path collective, the only RCCL traffic is the barrier / max-over-ranks of the timing and the final
PSNR reduction.  Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
PEAK_HBM_GBPS = 8000.0

ARCHS = {
    'GuidedResUnet': dict(name='GuidedResUnet', guided=True, in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
    'UNetSeeInDark': dict(name='UNetSeeInDark', in_nc=4, out_nc=4, nf=32, nframes=1, res=True, norm=True),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="frames per step and GPU (0 = default: 24 for 'once', 12 for 'iter', the batch size with --batch)")
    ap.add_argument("--mode", default="once", choices=["once", "iter"])
    ap.add_argument("--arch", default="GuidedResUnet", choices=list(ARCHS))
    ap.add_argument("--cfg", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config: 2 (headline) one 3000x4000 frame at a time, SNR-Net; 3 SIDD-validation-shaped items "
                         "(YOND_SIDD.py eval: 3000x5328 frame for the round-1 estimate, 32 blocks of 256x256, pipeline 'iter', batch-32 "
                         "forwards, block metrics); 4 UNetSeeInDark, batch 8 of 3000x4000 in one forward; 5 low light, no black-level "
                         "clip, 4000x6000, fp16 MFMA conv path")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0, help="frames per batched forward (cfg 4: 8)")
    ap.add_argument("--weights", default="denoising", choices=["denoising", "procedural"],
                    help="denoising: synthetic.denoising_state_dict (a real, weak denoiser: round 2 of 'iter' runs, PSNR is "
                         "meaningful); procedural: seeded random weights (round 2 ends at the reference's beta1 < 0 guard)")
    ap.add_argument("--distinct-frames", type=int, default=3, help="resident frames cycled through (different noise seeds)")
    ap.add_argument("--min-warmup-s", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (no sequential / fp32-mfma / parity legs)")
    ap.add_argument("--group", type=int, default=4, help="cfg 3: SIDD images denoised together (one batch-(32 x group) forward per round); 1 = one image at a time")
    ap.add_argument("--lanes", type=int, default=0, help="A/B: HIP streams the network passes of consecutive frames alternate between (pipeline.STREAM_LANES; 0 = its default)")
    ap.add_argument("--lane-pipelines", type=int, default=-1, help="A/B: 1 = every frame's whole chain on one lane (pipeline.LANE_PIPELINES / _ONCE), 0 = estimates on the side stream")
    ap.add_argument("--sequential", action="store_true", help="time one frame at a time (IterDenoise) instead of the two-stream driver")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiments: no HIP events around the kernels (no roofline objects)")
    ap.add_argument("--precision", default=None, choices=["fp32", "fp32-mfma", "fp16"],
                    help="fp32 (headline): fp32 results, 3x3 convolutions as fp32-accurate split-operand products on the fp16 MFMA; "
                         "fp32-mfma: every convolution on the fp32-input MFMA; fp16: BASELINE cfg 5 (not the headline configuration)")
    a = ap.parse_args(argv)
    if a.cfg == 3:
        a.mode = "iter"                                # the shipped SIDD runfile (runfiles/YOND/SIDD_simple+full_pre_grumix.yml)
        a.height, a.width = 256, 8192                  # the 32 blocks of an item side by side: the denoised Bayer pixels
        a.frames_per_step = a.frames_per_step or 16         # (four groups of four images: the two-stream driver overlaps consecutive groups)
        a.no_extras = True
    if a.cfg == 4:
        a.arch = 'UNetSeeInDark'
        a.batch = a.batch or 8
    if a.cfg == 5:
        a.precision = a.precision or 'fp16'
        a.height, a.width = a.height or 4000, a.width or 6000
    a.precision = a.precision or 'fp32'
    a.height, a.width = a.height or 3000, a.width or 4000
    if not a.frames_per_step:
        a.frames_per_step = 3 * a.batch if a.batch else (24 if a.mode == "once" else 12)     # (batches: three per step -- the stream driver overlaps consecutive batches)
        if a.cfg == 5:
            a.frames_per_step = 12
    if a.batch and a.frames_per_step % a.batch:
        ap.error("--frames-per-step must be a multiple of --batch")
    return a


def spawn_ranks(a, argv):
    """--gpus N without a torchrun environment: start the N ranks as a child job and relay its output.  Runs before any
    HIP call of this process (never re-exec a process that has initialised the GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"bench.py: starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


class GpuSampler(threading.Thread):
    """Clock and power of the busiest card while the timed region runs, sampled every 50 ms from sysfs hwmon
    (freq1_input: current shader clock in Hz; power1_average / power1_input: socket power in microwatts).  The clock the
    kernels really see is measured in-kernel by yond_clock_probe (`gfx_clock.in_kernel_mhz`): hwmon reads a little above it
    (MI355X_MICROARCH.md 'DVFS give-back'), and the DPM level table (pp_dpm_sclk) only names the ceiling."""

    def __init__(self):
        super().__init__(daemon=True)
        self.cards = []
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            f = os.path.join(hw, "freq1_input")
            pw = [q for q in (os.path.join(hw, "power1_average"), os.path.join(hw, "power1_input")) if os.path.exists(q)]
            if os.path.exists(f):
                self.cards.append((f, pw[0] if pw else None))
        self.mhz = {c[0]: [] for c in self.cards}
        self.watt = {c[0]: [] for c in self.cards}
        self.stop_flag = False

    def run(self):
        while not self.stop_flag and self.cards:
            for f, pw in self.cards:
                try:
                    self.mhz[f].append(float(open(f).read()) / 1e6)
                    if pw:
                        self.watt[f].append(float(open(pw).read()) / 1e6)
                except Exception:
                    pass
            time.sleep(0.05)

    def result(self):
        self.stop_flag = True
        best = max((f for f in self.mhz if self.watt[f] or self.mhz[f]), key=lambda f: (sum(self.watt[f]) / len(self.watt[f])) if self.watt[f]
                   else (sum(self.mhz[f]) / max(len(self.mhz[f]), 1)), default=None)
        if best is None:
            return {}
        out = {"hwmon_mhz": round(sum(self.mhz[best]) / max(len(self.mhz[best]), 1), 1), "samples": len(self.mhz[best])}
        if self.watt[best]:
            out["socket_power_w"] = round(sum(self.watt[best]) / len(self.watt[best]), 1)
        # every card of the node (an N-GPU run explains its own spread: the pool's cards hold 1.85-2.06 GHz under this load)
        out["per_card"] = [{"card": f.split("/")[4], "hwmon_mhz": round(sum(self.mhz[f]) / max(len(self.mhz[f]), 1), 1),
                            "socket_power_w": round(sum(self.watt[f]) / len(self.watt[f]), 1) if self.watt[f] else None}
                           for f in sorted(self.mhz) if self.mhz[f]]
        return out


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands out a
    share of its host -- 16 cores per GPU on this pool -- while os.cpu_count() reports the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline_and_parity(a, arch, dev, net_factory):
    """The oracle (CPU restatement of the reference path; test infrastructure used ONLY as the reported baseline and as
    the checker of the parity leg) on a bounded sample of the workload, at 1 thread (the reference pins 1 thread,
    utils/utils.py:2-6) and on all host cores; then the HIP path on the same sample frame and weights, compared with the
    oracle's output."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import yond_oracle as O
    from yond_public_amd import pipeline as P
    H, W = (a.height, a.width) if a.cfg in (2, 3) else (2048, 3072)      # the headline's own 3000x4000 frame (bounded sample: one frame)
    noisy, clean = O.synth_noisy(H, W, 4.0, 6.0, 100 if a.cfg == 3 else 0, clip=(a.cfg != 5))
    sd = O.denoising_state_dict(arch, 0) if a.weights == "denoising" else O.procedural_state_dict(arch, 0)
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': a.mode, 'max_iter': 1, 'full_dn': True,
            'collab_sidd256': False}
    lr_full = None
    if a.cfg == 3:                                  # one SIDD-shaped item: [32][256][256] blocks + the full frame of the round-1 estimate
        pipe = dict(SIDD_PIPE)
        lr_full, _ = O.synth_noisy(SIDD_FULL[0], SIDD_FULL[1], 4.0, 6.0, 500)
        noisy = np.array(np.split(noisy, 32, axis=-1))
    old = torch.get_num_threads()
    ncores = host_cores()
    runs = {}
    for threads in (1, ncores):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        ref = O.IterDenoise(noisy, arch, sd, pipe, lr_full=lr_full)
        runs[threads] = time.perf_counter() - t0
    torch.set_num_threads(old)
    what = (f"one {H}x{W} synthetic Bayer frame, same pipeline ('{a.mode}', {len(ref['raw_dns'])} pass(es)) and weights, "
            f"oracle/yond_oracle.py (NumPy/SciPy + PyTorch-CPU)")
    if a.cfg == 3:
        what = (f"one SIDD-shaped item ({SIDD_FULL[0]}x{SIDD_FULL[1]} estimate frame + 32 blocks of 256x256), IterDenoise of the shipped SIDD "
                f"pipeline ({len(ref['raw_dns'])} passes, 64 batch-1 forwards as in the reference), oracle/yond_oracle.py")
    base = {"value": round(H * W / 1e6 / runs[1], 4), "unit": "Bayer MP/s", "cores": 1, "kind": "port",
            "sample": f"{what}, {runs[1]:.1f} s",
            "all_cores": {"value": round(H * W / 1e6 / runs[ncores], 4), "unit": "Bayer MP/s", "cores": ncores,
                          "sample": f"the same, torch.set_num_threads({ncores}), {runs[ncores]:.1f} s"}}
    net = net_factory(a.precision)
    res = P.IterDenoise(torch.from_numpy(noisy).to(dev), net, arch, pipe, lr_full=None if lr_full is None else torch.from_numpy(lr_full).to(dev))
    parity = {"sample": f"{H}x{W} frame of cpu_baseline", "passes": len(res['raw_dns'])}
    if len(res['raw_dns']) != len(ref['raw_dns']):
        parity["error"] = f"HIP ran {len(res['raw_dns'])} passes, oracle {len(ref['raw_dns'])}"
    else:
        got = res['raw_dns'][-1].cpu().numpy().astype(np.float64)
        want = np.asarray(ref['raw_dns'][-1], np.float64)
        mse = float(np.mean((got - want) ** 2))
        parity.update({
            "max_abs_vs_oracle": float(np.max(np.abs(got - want))),
            "psnr_hip_vs_oracle_db": round(10 * np.log10(1.0 / max(mse, 1e-30)), 2),
            "psnr_vs_clean_db": {"hip": round(O.psnr(got, clean), 4), "oracle": round(O.psnr(want, clean), 4)},
            "psnr_delta_vs_oracle_db": round(O.psnr(got, clean) - O.psnr(want, clean), 6),
            "regs_hip": [[float(v) for v in r] for r in res['regs']],
            "regs_oracle": [[float(v) for v in r] for r in ref['regs']]})
    return base, parity


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE collected separately, gfx950 corrections applied by tools/pmc_traffic.py); None if not collected."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            t = json.load(open(path)).get(kernel)
            if t:
                return {"hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "unit": "B", "source": "profiles/" + os.path.basename(path)}
        except Exception:
            pass
    return None


SIDD_FULL = (3000, 5328)          # SURVEY 8d cfg 3: synthetic stand-ins of the SIDD full frames
SIDD_PIPE = {'data_type': 'SIDD', 'full_est': True, 'est_type': 'simple+full', 'k': 29, 'full_dn': False, 'vst_type': 'exact',
             'bias_corr': 'pre', 'denoiser_type': 'gru32n', 'iter': 'iter', 'max_iter': 1, 'clip': False}     # the shipped runfile's `pipeline`
REF_SIDD_S_PER_IMAGE = 5.1        # logs/log_YOND_SIDD_simple+full_pre_grumix_iter.log:10,130 (204 s / 40 images; other hardware: context only)


def sidd_items(n, dev, rank=0):
    """n SIDD-validation-shaped synthetic items resident on the device (reference: YOND_SIDD.py:507-514 hands IterDenoise
    `lr` [32][256][256], the ground truth and the full-resolution frame the round-1 estimate is taken from)."""
    import numpy as np
    import torch
    from yond_public_amd import synthetic as S
    items = []
    for k in range(n):
        noisy, clean = S.synth_noisy(256, 8192, 4.0, 6.0, 100 + 1000 * rank + k)
        full, _ = S.synth_noisy(SIDD_FULL[0], SIDD_FULL[1], 4.0, 6.0, 500 + 1000 * rank + k)
        items.append({'lr': torch.from_numpy(np.array(np.split(noisy, 32, axis=-1))).to(dev), 'hr': torch.from_numpy(clean).to(dev),
                      'lr_full': torch.from_numpy(full).to(dev)})
    return items


def leg_roofline(plan, run_once, sync, fp16_operands=False):
    """The dominant 3x3 stride-1 convolution kernel of one more, instrumented pass of a bench leg: HIP event pairs around every such
    launch of `run_once()`; algorithmic FLOPs per launch / mean launch time against the peak of the MFMA the kernel issues."""
    is33 = lambda t: t.startswith("conv_mfma_kernel<3,1") or t.startswith("conv_wino_kernel") or t.startswith("conv_split_kernel<1,")
    old = (getattr(plan, 'prof', None), getattr(plan, 'prof_only', None), getattr(plan, 'prof_every', 1))
    plan.prof, plan.prof_only, plan.prof_every = [], is33, 1
    try:
        run_once()
        sync()
        prof = plan.prof
    finally:
        plan.prof, plan.prof_only, plan.prof_every = old
    per = {}
    for tag, flops, e0, e1 in prof:
        k = per.setdefault(tag, [0, 0.0, 0.0])
        k[0] += 1
        k[1] += e0.elapsed_time(e1)
        k[2] += flops
    dom = max((t for t in per if is33(t)), key=lambda t: per[t][1], default=None)
    if dom is None:
        return None
    n, ms, fl = per[dom]
    peak = PEAK_F16_MFMA_TFLOPS if (fp16_operands or dom.startswith("conv_split_kernel")) else PEAK_F32_MFMA_TFLOPS
    ach = fl / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "launches": n, "avg_launch_ms": round(ms / n, 4), "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3),
            "measured": "HIP events around every 3x3 stride-1 launch of one instrumented pass behind the leg's timed loop"}


def sidd_eval_item(item, net, arch, P):
    """One image of YOND_SIDD.eval (:507-536): IterDenoise (full-frame self NLE, bias LUT, 32 blocks as ONE batch-32 forward,
    collaborative NLE with the SIDD_256 re-tiling, second pass) + per-block PSNR / SSIM of both rounds on the device."""
    res = P.IterDenoise(item['lr'], net, arch, SIDD_PIPE, lr_full=item['lr_full'])
    if len(res['raw_dns']) != 2:
        raise SystemExit(f"bench.py: the SIDD pipeline ran {len(res['raw_dns'])} pass(es), expected 2 (regs {res['regs']})")
    res['metrics'] = [P.block_metrics(dn, item['hr']) for dn in res['raw_dns']]
    return res


def sidd_eval_group(items, net, arch, P):
    """G images of YOND_SIDD.eval as one group (yond_public_amd/YOND_SIDD.py eval, --group): round 1 of the G images is ONE batch-(32 G)
    forward, round 2 another; estimates, tables, t and the block metrics stay per image (per image the results are sidd_eval_item's: tests/test_hip_eval.py)."""
    ress = P.IterDenoiseGroup([(it['lr'], it['lr_full']) for it in items], net, arch, SIDD_PIPE)
    for res, it in zip(ress, items):
        if len(res['raw_dns']) != 2:
            raise SystemExit(f"bench.py: the SIDD pipeline ran {len(res['raw_dns'])} pass(es), expected 2 (regs {res['regs']})")
        res['metrics'] = [P.block_metrics(dn, it['hr']) for dn in res['raw_dns']]
    return ress


def sidd_eval_stream(items, group, net, arch, P):
    """YOND_SIDD.eval's loop over `items` as the driver runs it: groups of `group` images, consecutive groups overlapped on two HIP streams
    (pipeline.denoise_stream_groups), the block metrics of a finished group on a third.  Returns the last image's result."""
    groups = [items[i:i + group] for i in range(0, len(items), group)]
    met = {}

    def finish(gi, ress):
        met[gi] = [[P.block_metrics(dn, it['hr']) for dn in res['raw_dns']] for it, res in zip(groups[gi], ress)]
    last = None
    for gi, ress in enumerate(P.denoise_stream_groups(([(it['lr'], it['lr_full']) for it in g] for g in groups), net, arch, SIDD_PIPE, finish=finish)):
        for res, m in zip(ress, met.pop(gi)):
            if len(res['raw_dns']) != 2:
                raise SystemExit(f"bench.py: the SIDD pipeline ran {len(res['raw_dns'])} pass(es), expected 2 (regs {res['regs']})")
            res['metrics'] = m
            last = res
    return last


def batch_stream(frames, B, net, arch, pipe, P):
    """cfg 4's loop: B frames per forward, the B estimators of batch k+1 on a second HIP stream under the batched network pass of batch k
    (pipeline.denoise_stream_batches).  Returns the LAST batch in IterDenoiseBatch's shape (raw_dns [[B][H][W]], regs [[B]], params [[B]])."""
    import torch
    last = []
    for r in P.denoise_stream_batches(frames, B, net, arch, pipe):
        last.append(r)
        if len(last) > B:
            last.pop(0)
    return dict(raw_dns=[torch.stack([r['raw_dns'][0] for r in last])], regs=[[r['regs'][0] for r in last]], params=[[r['params'][0] for r in last]])


def timed_region(run_step, steps, sync, D, dev):
    """EXACTLY `steps` steps bracketed by barrier + device synchronisation on both sides; returns (the MAX over ranks of the
    elapsed time, every rank's time for its own K steps -- taken before the closing barrier -- in rank order)."""
    sync()
    D.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    sync()
    own = time.perf_counter() - t0                   # this rank's own K steps (before it waits for the others)
    D.barrier()
    sync()
    local = time.perf_counter() - t0
    return D.max_over_ranks(local, dev), D.gather_over_ranks(own, dev)


def check_world(a, D):
    """The process group's own world size must be the N of --gpus (a job launched with another --nproc-per-node, or ranks
    that did not all join, would otherwise report an aggregate over the wrong number of GPUs)."""
    gw = D.group_world_size()
    if gw is not None and gw != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the process group ({D.STATS['backend']}) has {gw} rank(s)")
    if gw is None and a.gpus != 1:
        raise SystemExit(f"bench.py: --gpus {a.gpus} without a process group")
    return gw


def static_notes():
    """NOT measured by the run that prints them: constants and provenance a reader of the line may want beside it."""
    return {"not_measured_by_this_run": True,
            "scaling": ("this line is one point; the 1/2/4/8-GPU curve is the driver's SCALE_rNN.json where it had a multi-GPU node.  The path "
                        "shards by images with no data-path collective (DESIGN section 6)"),
            "bare_loop_ceiling_of_split_products": {
                "issued_frac": 0.60, "algorithmic_frac": 0.20, "in_kernel_mhz": 1690,
                "what": ("a bare loop of the three MFMAs per fp32 product block and their LDS fragment reads (no global memory, no barriers) "
                         "under the board's power cap, measured ONCE by a probe"),
                "source": "tools/probe/mfma_shape_probe.hip, profiles/r05_experiments/mfma_shape_probe.txt"}}


def stub_main(a):
    """Tests only (YOND_BENCH_STUB=1; tests/test_bench_launcher.py): the launcher, the rendezvous, the timed-region protocol
    (barriers, max over ranks, per-rank gather) and rank 0's line with the hot path replaced by a sleep, on gloo -- so that the
    N > 1 plumbing of `bench.py --gpus N` runs on a CPU-only box.  Never a measurement: the line says so."""
    from yond_public_amd import distributed as D
    rank, local, world = D.init(backend="gloo")
    gw = check_world(a, D)
    step_s = float(os.environ.get("YOND_BENCH_STUB_STEP_S", "0.01")) * (1 + rank)      # (uneven ranks: the max must win)
    elapsed, per_rank = timed_region(lambda: time.sleep(step_s), a.steps, lambda: None, D, None)
    if os.environ.get("YOND_BENCH_STUB_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    if rank == 0:
        mp = a.height * a.width / 1e6
        n_timed = a.steps * a.frames_per_step
        print(json.dumps({"metric": "STUB (no GPU work: launcher / rendezvous / reduction plumbing only)", "value": round(world * n_timed * mp / elapsed, 2),
                          "unit": "Bayer MP/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
                          "data": "stub", "static_notes": static_notes(), "collectives": dict(D.STATS, world_size=gw,
                                                               per_rank_mp_per_s=[round(n_timed * mp / t, 2) for t in per_rank],
                                                               per_rank=[{"rank": r, "mp_per_s": round(n_timed * mp / t, 2), "in_kernel_mhz": None}
                                                                         for r, t in enumerate(per_rank)])}), flush=True)
    D.barrier()
    D.finalize()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(a, argv))
    if env_world is not None and int(env_world) != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={env_world}; launch with --nproc-per-node {a.gpus}")
    if os.environ.get("YOND_BENCH_STUB") == "1":
        return stub_main(a)

    import numpy as np
    import torch
    from yond_public_amd import distributed as D
    from yond_public_amd import pipeline as P
    from yond_public_amd import synthetic as S
    from yond_public_amd import archs as A
    from yond_public_amd import _lib as L

    rank, local, world = D.init()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    if a.lanes:
        P.STREAM_LANES = a.lanes
    if a.lane_pipelines >= 0:
        P.LANE_PIPELINES = P.LANE_PIPELINES_ONCE = bool(a.lane_pipelines)
    group_world = check_world(a, D)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    L.load()

    arch = ARCHS[a.arch]

    def make_net(precision):
        net = getattr(A, arch['name'])(dict(arch, precision=precision))
        net.load_state_dict(S.denoising_state_dict(net, 0) if a.weights == "denoising" else S.procedural_state_dict(net, 0))
        return net.to(dev).eval()

    net = make_net(a.precision)
    H, W = a.height, a.width
    clip = a.cfg != 5                                     # cfg 5: no black-level clip (negative DN reach the VST)
    expo = 0.2 if a.cfg == 5 else 1.0                     # cfg 5: low light
    frames, cleans = [], []
    if a.cfg == 3:
        frames = sidd_items(max(1, min(a.distinct_frames, 2)), dev, rank)
        cleans = [it['hr'] for it in frames]
    for i in range(max(1, a.distinct_frames) if a.cfg != 3 else 0):
        if a.cfg == 5:
            rng = np.random.default_rng(1997 + 1000 * rank + i)
            clean = (S.synth_clean(H, W) * expo).astype(np.float32)
            noisy = ((rng.poisson(clean * 959.0 / 4.0) * 4.0 + rng.normal(0.0, 25.0, clean.shape)) / 959.0).astype(np.float32)
        else:
            noisy, clean = S.synth_noisy(H, W, 4.0, 6.0, 1000 * rank + i, clip=clip)
        frames.append(torch.from_numpy(noisy).to(dev))
        cleans.append(torch.from_numpy(clean).to(dev))
    pipe = {'k': 29, 'vst_type': 'exact', 'bias_corr': 'pre', 'iter': a.mode, 'max_iter': 1, 'full_dn': True,
            'collab_sidd256': False}
    n_pass = 2 if a.mode == "iter" else 1
    stream_driver = not a.sequential and not a.batch and a.cfg != 3
    F = a.frames_per_step

    def one(frame, netx=None):
        if a.cfg == 3:
            return sidd_eval_item(frame, netx or net, arch, P)
        res = P.IterDenoise(frame, netx or net, arch, pipe)
        if len(res['raw_dns']) != n_pass:
            raise SystemExit(f"bench.py: pipeline '{a.mode}' ran {len(res['raw_dns'])} pass(es), expected {n_pass} "
                             f"(regs {res['regs']}): with --weights procedural the reference's guard ends round 2")
        return res

    def run(nframes, sequential=False):
        """nframes passes of the hot path over the resident frames (cycled)."""
        last = None
        if a.batch and not sequential and not a.sequential and a.mode == "once":
            last = batch_stream((frames[i % len(frames)] for i in range(nframes // a.batch * a.batch)), a.batch, net, arch, pipe, P)
        elif a.batch:
            for b in range(nframes // a.batch):
                last = P.IterDenoiseBatch([frames[(b * a.batch + j) % len(frames)] for j in range(a.batch)], net, arch, pipe)
        elif stream_driver and not sequential:
            for last in P.denoise_stream((frames[i % len(frames)] for i in range(nframes)), net, arch, pipe):
                pass
        elif a.cfg == 3 and a.group > 1 and not sequential and not a.sequential:
            last = sidd_eval_stream([frames[i % len(frames)] for i in range(nframes)], a.group, net, arch, P)
        elif a.cfg == 3 and a.group > 1:
            for i in range(0, nframes, a.group):
                last = sidd_eval_group([frames[(i + j) % len(frames)] for j in range(min(a.group, nframes - i))], net, arch, P)[-1]
        else:
            for i in range(nframes):
                last = one(frames[i % len(frames)])
        return last

    # warm-up: W steps, continued until --min-warmup-s has passed (clocks and caches settle)
    t_w = time.perf_counter()
    done = 0
    while done < a.warmup or (time.perf_counter() - t_w < a.min_warmup_s and done < 50 * max(a.warmup, 1)):
        res = run(F)
        torch.cuda.synchronize()
        done += 1
    warm_s = time.perf_counter() - t_w
    plan = P._plan_of(net, dev)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    # HIP events (on the launch stream) bracket the launches of the 3x3 stride-1 convolution kernels -- the dominant
    # family: 18 launches per forward, 91 % of the MACs -- in every 4th forward of the timed region; an event pair costs
    # the stream a few microseconds (reported: `without_kernel_events`), so the other kernels and the VST / NLE stages are
    # timed in one more, instrumented pass behind it
    is33 = lambda t: t.startswith("conv_mfma_kernel<3,1") or t.startswith("conv_wino_kernel") or t.startswith("conv_split_kernel<1,")
    plan.prof = None if a.no_kernel_events else []
    plan.prof_only = is33
    plan.prof_every = 4                  # every 4th forward carries the event pairs (their cost is reported: `without_kernel_events`)
    sclk = GpuSampler() if rank == 0 else None
    if sclk:
        sclk.start()
    # in-kernel shader clock of the split-operand convolution launches of the timed region: workgroup 0 of every launch adds its
    # elapsed shader cycles and 100 MHz reference ticks to two device counters (YondConvDesc.clk).  (A probe kernel BESIDE the
    # region cannot be used: these kernels take every vector register of their CUs, so a single foreign wave keeps one of the
    # 256 persistent workgroups waiting for a CU -- measured: every launch 1.45x longer.)
    plan.clk = torch.zeros(2, dtype=torch.int64, device=dev)
    last_res = {}

    def run_step():
        last_res['res'] = run(F)
    elapsed, per_rank_s = timed_region(run_step, a.steps, torch.cuda.synchronize, D, dev)
    res = last_res['res']
    sclk_res = sclk.result() if sclk else None
    clk_c, clk_r = (int(v) for v in plan.clk.cpu())
    plan.clk = None
    per_rank_mhz = D.gather_over_ranks(clk_c / clk_r * 100.0 if clk_r > 0 else 0.0, dev)      # every rank's own in-kernel clock
    if rank == 0:
        sclk_res = dict(sclk_res or {}, in_kernel_mhz=round(clk_c / clk_r * 100.0, 1) if clk_r > 0 else None,
                        source="in_kernel_mhz: s_memtime / s_memrealtime of workgroup 0 of every split-operand convolution launch of the "
                               "timed region (YondConvDesc.clk); hwmon_mhz / socket_power_w: sysfs hwmon of the busiest card, 50 ms "
                               "samples during the timed region")
    prof, plan.prof = plan.prof or [], None
    # With the network passes of two frames on two lanes (pipeline.STREAM_LANES) the persistent workgroups of one lane's launch take the CUs as the other
    # lane's launch retires them: a launch's wall duration between its events then includes its wait for the other lane's workgroups and is no longer its
    # cost.  The roofline object therefore comes from the SAME job run on one lane right behind the timed region (>= 2 steps, the same event pairs); the
    # timed region's own wall durations stay beside it (`timed_region_wall`).
    prof_wall = None
    if prof and P.STREAM_LANES > 1 and not a.sequential:
        prof_wall = prof
        lanes_was, P.STREAM_LANES = P.STREAM_LANES, 1
        try:
            run(F)                                          # (lane 0 alone: its clocks and caches as they will be measured)
            torch.cuda.synchronize()
            plan.prof = []
            t_r = time.perf_counter()
            n_roof = 0
            while n_roof < 2 or time.perf_counter() - t_r < 0.5:
                run(F)
                n_roof += 1
            torch.cuda.synchronize()
            one_lane_ms_per_frame = (time.perf_counter() - t_r) / (n_roof * F) * 1e3
        finally:
            P.STREAM_LANES = lanes_was
        prof, plan.prof = plan.prof or [], None
    plan.prof_only = None
    plan.prof_every = 1
    n_timed = a.steps * F
    last_frame = frames[(n_timed - 1) % len(frames)]
    last_clean = cleans[(n_timed - 1) % len(frames)]
    last_noisy = torch.cat(list(last_frame['lr']), dim=-1) if a.cfg == 3 else last_frame

    # one frame at a time (wall time from 'noisy frame resident' to 'denoised frame resident')
    seq = None
    if stream_driver and not a.no_extras:
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        n_seq = 0
        while n_seq < 20 or time.perf_counter() - t1 < 1.0:
            one(frames[n_seq % len(frames)])
            torch.cuda.synchronize()
            n_seq += 1
        el = D.max_over_ranks(time.perf_counter() - t1, dev)
        seq = {"value": round(world * n_seq * H * W / 1e6 / el, 2), "unit": "Bayer MP/s", "ms_per_frame": round(el / n_seq * 1e3, 3),
               "frames": n_seq, "definition": "IterDenoise one frame at a time, synchronised after every frame"}

    # what the HIP event pairs inside the timed region cost: the same job for >= 1 s without them
    noev = None
    if not a.no_extras and not a.no_kernel_events:
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        n_ne = 0
        while n_ne < 2 or time.perf_counter() - t1 < 1.0:
            run(F)
            torch.cuda.synchronize()
            n_ne += 1
        el = D.max_over_ranks(time.perf_counter() - t1, dev)
        noev = {"value": round(world * n_ne * F * H * W / 1e6 / el, 2), "unit": "Bayer MP/s", "ms_per_frame": round(el / (n_ne * F) * 1e3, 3),
                "steps": n_ne, "definition": "the timed region's job without the HIP event pairs around the 3x3 stride-1 launches"}

    # the shipped default pipeline: 'iter' (two rounds per frame, YOND_SIDD.py:419-472), one frame at a time, >= 1 s
    iter_leg = None
    if a.mode == "once" and not a.no_extras and not a.batch:
        pipe_it = dict(pipe, iter='iter')
        r_it = P.IterDenoise(frames[0], net, arch, pipe_it)
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        n_it = 0
        while n_it < 10 or time.perf_counter() - t1 < 1.0:
            r_it = P.IterDenoise(frames[n_it % len(frames)], net, arch, pipe_it)
            torch.cuda.synchronize()
            n_it += 1
        el = D.max_over_ranks(time.perf_counter() - t1, dev)
        if len(r_it['raw_dns']) != 2:
            raise SystemExit(f"bench.py: the 'iter' leg ran {len(r_it['raw_dns'])} pass(es), expected 2 (regs {r_it['regs']})")
        seq_ms = el / n_it * 1e3
        # ... and on the two-stream driver (pipeline._denoise_stream_chain_iter: frame k's collaborative estimate under another frame's network pass)
        for _ in P.denoise_stream((frames[i % len(frames)] for i in range(4)), net, arch, pipe_it):
            pass
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        n_st = 24
        for r_it in P.denoise_stream((frames[i % len(frames)] for i in range(n_st)), net, arch, pipe_it):
            pass
        torch.cuda.synchronize()
        el = D.max_over_ranks(time.perf_counter() - t1, dev)
        if len(r_it['raw_dns']) != 2:
            raise SystemExit(f"bench.py: the streamed 'iter' leg ran {len(r_it['raw_dns'])} pass(es), expected 2 (regs {r_it['regs']})")
        iter_leg = {"value": round(world * n_st * H * W / 1e6 / el, 2), "unit": "Bayer MP/s", "ms_per_frame": round(el / n_st * 1e3, 3),
                    "frames": n_st, "passes_per_frame": 2,
                    "one_frame_at_a_time_ms_per_frame": round(seq_ms, 3),
                    "definition": "pipeline 'iter' (self NLE + denoise, collaborative NLE + denoise: the reference's shipped default) through denoise_stream: frame k's "
                                  "estimates on a second HIP stream under other frames' network passes; `one_frame_at_a_time_ms_per_frame`: IterDenoise, synchronised after every frame"}

    stage_prof, prof_all = [], []
    if not a.no_kernel_events:
        passes = []                                  # three instrumented passes, the one with the median stage total is reported (a pass whose stage
        for _ in range(3):                           # holds an allocator refill -- a host gap between two event records -- is not the stage's time)
            plan.prof = []
            P.PROF = []
            if a.batch:
                run(a.batch)
            else:
                one(last_frame)
            torch.cuda.synchronize()
            passes.append((sum(e0.elapsed_time(e1) for _, e0, e1 in P.PROF), plan.prof, P.PROF))
            plan.prof, P.PROF = None, None
        _, prof_all, stage_prof = sorted(passes, key=lambda t: t[0])[1]
    inst_frames = a.batch or 1

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
        if prof_wall is not None:
            wall = [e0.elapsed_time(e1) for tag, _, e0, e1 in prof_wall if tag == dom]
            roof["measured"] = (f"HIP events on the launch stream around the 3x3 stride-1 launches of every 4th forward while the timed region's job ({F} frames per step, {n_roof} steps) ran on ONE "
                                f"lane right behind the timed region; rocprofv3 counterpart: profiles/r06_bench_{'cfg5' if a.cfg == 5 else 'once'}_one_lane_kernel_stats.csv (bench.py --lanes 1)")
            roof["one_lane_ms_per_frame"] = round(one_lane_ms_per_frame, 3)
            roof["timed_region_wall"] = {"launches": len(wall), "avg_launch_ms": round(sum(wall) / max(len(wall), 1), 4),
                                         "note": f"the timed region runs the network passes of consecutive frames on {P.STREAM_LANES} lanes: the persistent workgroups of one lane's launch take "
                                                 "the CUs as the other lane's launch retires them, so the time between a launch's HIP events there includes its wait in the hardware queue "
                                                 "for the other lane's workgroups; rocprofv3's begin / end timestamps of the default command -- the launch's own execution -- stay within ~3 % of "
                                                 f"the one-lane durations (profiles/r06_bench_{'cfg5' if a.cfg == 5 else 'once'}_kernel_stats.csv beside ..._one_lane_kernel_stats.csv)"}
        if dom.startswith("conv_wino_kernel"):
            # Winograd F(2x2,3x3) issues 16 multiplications per patch where the direct algorithm has 36: `achieved`
            # counts the ALGORITHMIC flops of the convolution; the flops the MFMA unit really executes are 16/36 of that
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
        tot_ms = sum(stage_ms.values()) / inst_frames
        bpp = 24.0 if a.mode == "once" else 56.0
        alg_bytes = bpp * H * W
        if a.cfg == 3:
            # self NLE on the full frame (8 B per pixel), two K1 + K4 passes and the collaborative NLE on the 32 blocks (16 + 16 + 16 B)
            alg_bytes = 8.0 * SIDD_FULL[0] * SIDD_FULL[1] + 48.0 * H * W
            bpp = alg_bytes / (H * W)
        gbs = alg_bytes / (tot_ms * 1e-3) / 1e9
        roof_hbm = {"stage": "VST+NLE (K1, K4, K5-K7)", "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBPS, 4), "algorithmic_bytes_per_bayer_px": bpp, "ms_per_frame": round(tot_ms, 4),
                    "stage_ms_per_frame": {k: round(v / inst_frames, 4) for k, v in stage_ms.items()},
                    "measured": "HIP events, the median of three instrumented passes after the timed region"}
    conv_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in prof_all) / inst_frames
    conv_fl = sum(fl for _, fl, _, _ in prof_all) / inst_frames

    # the same job with every convolution on the fp32-input MFMA (Winograd / direct kernels), >= 1 s, beside the headline
    strict = None
    if a.precision == "fp32" and not a.no_extras and not a.batch:
        net2 = make_net('fp32-mfma')
        r2 = one(last_frame, net2)
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        n2 = 0
        while n2 < 10 or time.perf_counter() - t1 < 1.0:
            r2 = one(frames[n2 % len(frames)], net2)
            torch.cuda.synchronize()
            n2 += 1
        el2 = D.max_over_ranks(time.perf_counter() - t1, dev)
        r2 = one(last_frame, net2)
        r1 = one(last_frame)
        dmax = (r2['raw_dns'][-1] - r1['raw_dns'][-1]).abs().max().item()
        strict = {"value": round(world * n2 * H * W / 1e6 / el2, 2), "unit": "Bayer MP/s", "ms_per_frame": round(el2 / n2 * 1e3, 3),
                  "frames": n2, "driver": "one frame at a time", "max_abs_output_difference_to_headline_path": dmax}
        del net2, r2, r1

    # BASELINE.json's two other GPU configurations on the same line (one GPU, rank 0, >= 1 s each): configs[3] = UNetSeeInDark,
    # batch 8 of 3000 x 4000 frames in ONE forward; configs[4] = 4000 x 6000 low-light frames without black-level clip on the
    # fp16 MFMA path.  (`--cfg 4` / `--cfg 5` run them as the timed region proper.)
    others = None
    if a.cfg == 2 and a.mode == "once" and a.precision == "fp32" and not a.no_extras and not a.batch and world == 1 and stream_driver:
        others = {}
        arch4 = ARCHS['UNetSeeInDark']
        net4 = getattr(A, arch4['name'])(dict(arch4, precision='fp32'))
        net4.load_state_dict(S.denoising_state_dict(net4, 0))
        net4 = net4.to(dev).eval()
        batch8 = [frames[j % len(frames)] for j in range(8)]
        batch_stream((batch8[j % 8] for j in range(16)), 8, net4, arch4, pipe, P)
        torch.cuda.synchronize()
        t4, n4 = time.perf_counter(), 0
        while n4 < 3 or time.perf_counter() - t4 < 1.0:
            r4 = batch_stream((batch8[j % 8] for j in range(24)), 8, net4, arch4, pipe, P)
            n4 += 3
        torch.cuda.synchronize()
        el4 = time.perf_counter() - t4
        P.IterDenoiseBatch(batch8, net4, arch4, pipe)
        torch.cuda.synchronize()
        t4b, n4b = time.perf_counter(), 0
        while n4b < 2 or time.perf_counter() - t4b < 0.5:
            P.IterDenoiseBatch(batch8, net4, arch4, pipe)
            torch.cuda.synchronize()
            n4b += 1
        el4b = time.perf_counter() - t4b
        others["cfg4_unet_batch8"] = {"value": round(n4 * 8 * H * W / 1e6 / el4, 2), "unit": "Bayer MP/s", "ms_per_frame": round(el4 / (n4 * 8) * 1e3, 3),
                                      "frames": n4 * 8, "one_batch_at_a_time_ms_per_frame": round(el4b / (n4b * 8) * 1e3, 3),
                                      "workload": f"configs[3]: UNetSeeInDark(nf=32), {H}x{W} frames, per-frame NLE, ONE batched forward of 8; the estimators of batch k+1 on a "
                                                  "second HIP stream under the forward of batch k (pipeline.denoise_stream_batches)"}
        try:
            others["cfg4_unet_batch8"]["roofline"] = leg_roofline(P._plan_of(net4, dev), lambda: P.IterDenoiseBatch(batch8, net4, arch4, pipe),
                                                                  torch.cuda.synchronize)
        except Exception as e:
            others["cfg4_unet_batch8"]["roofline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        del net4, r4, batch8
        torch.cuda.empty_cache()
        # configs[2] (the reference's only shipped entry point, YOND_SIDD.py eval): per image the full-frame self NLE, two rounds of
        # batch-32 forwards over the 32 blocks, the collaborative NLE, per-block PSNR / SSIM of both rounds (`--cfg 3` times it as
        # the region proper; the reference's log shows 5.1 s per image on its authors' GPU)
        try:
            it3 = sidd_items(2, dev)
            G3 = max(1, a.group)
            grp3 = [it3[j % 2] for j in range(G3)]
            sidd_eval_stream([it3[j % 2] for j in range(4 * G3)], G3, net, arch, P)
            torch.cuda.synchronize()
            t3, n3 = time.perf_counter(), 0
            while n3 < 8 or time.perf_counter() - t3 < 1.0:
                r3 = sidd_eval_stream([it3[j % 2] for j in range(8 * G3)], G3, net, arch, P)
                n3 += 8 * G3
            torch.cuda.synchronize()
            el3 = time.perf_counter() - t3
            # ... one group at a time (synchronised after every group)
            sidd_eval_group(grp3, net, arch, P)
            torch.cuda.synchronize()
            t3g, n3g = time.perf_counter(), 0
            while n3g < 8 or time.perf_counter() - t3g < 0.5:
                sidd_eval_group(grp3, net, arch, P)
                n3g += G3
            torch.cuda.synchronize()
            el3g = time.perf_counter() - t3g
            # ... and one image at a time (the reference's loop shape; round 5's number)
            sidd_eval_item(it3[0], net, arch, P)
            torch.cuda.synchronize()
            t31, n31 = time.perf_counter(), 0
            while n31 < 4 or time.perf_counter() - t31 < 0.5:
                sidd_eval_item(it3[n31 % 2], net, arch, P)
                n31 += 1
            torch.cuda.synchronize()
            el31 = time.perf_counter() - t31
            others["cfg3_sidd_eval"] = {"images_per_s": round(n3 / el3, 2), "ms_per_image": round(el3 / n3 * 1e3, 3),
                                        "value": round(n3 * 256 * 8192 / 1e6 / el3, 2), "unit": "Bayer MP/s (denoised blocks: 2.097 MP per image)",
                                        "full_frame_mp_per_s": round(n3 * SIDD_FULL[0] * SIDD_FULL[1] / 1e6 / el3, 1), "images": n3,
                                        "psnr_iter0_iter1": [round(float(np.mean(m[0])), 3) for m in r3['metrics']],
                                        "reference_s_per_image": REF_SIDD_S_PER_IMAGE,
                                        "workload": f"configs[2]: SIDD-shaped synthetic items ({SIDD_FULL[0]}x{SIDD_FULL[1]} estimate frame + 32 blocks of 256x256), "
                                                    f"YOND_SIDD.eval's loop body (IterDenoise 'iter' + block metrics), {G3} images per group: round 1 of a group is ONE "
                                                    f"batch-{32 * G3} forward, round 2 another; estimates / tables / metrics per image; consecutive groups overlapped on two HIP streams "
                                                    "(pipeline.denoise_stream_groups), as yond_public_amd/YOND_SIDD.py eval runs them",
                                        "group": G3, "one_group_at_a_time_ms_per_image": round(el3g / n3g * 1e3, 3), "one_image_at_a_time_ms_per_image": round(el31 / n31 * 1e3, 3)}
            others["cfg3_sidd_eval"]["roofline"] = leg_roofline(plan, lambda: sidd_eval_group(grp3, net, arch, P), torch.cuda.synchronize)
            del it3, r3
        except Exception as e:
            others["cfg3_sidd_eval"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()
        H5, W5 = 4000, 6000
        rng5 = np.random.default_rng(1997)
        clean5 = (S.synth_clean(H5, W5) * 0.2).astype(np.float32)
        noisy5 = ((rng5.poisson(clean5 * 959.0 / 4.0) * 4.0 + rng5.normal(0.0, 25.0, clean5.shape)) / 959.0).astype(np.float32)
        f5 = torch.from_numpy(noisy5).to(dev)
        del clean5, noisy5
        net5 = make_net('fp16')
        for _ in P.denoise_stream((f5 for _ in range(2)), net5, arch, pipe):
            pass
        torch.cuda.synchronize()
        t5, n5 = time.perf_counter(), 0
        while n5 < 12 or time.perf_counter() - t5 < 1.0:
            for _ in P.denoise_stream((f5 for _ in range(6)), net5, arch, pipe):
                pass
            torch.cuda.synchronize()
            n5 += 6
        el5 = time.perf_counter() - t5
        others["cfg5_fp16_4000x6000"] = {"value": round(n5 * H5 * W5 / 1e6 / el5, 2), "unit": "Bayer MP/s", "ms_per_frame": round(el5 / n5 * 1e3, 3),
                                         "frames": n5, "dtype": "f16 MFMA operands, f32 accumulate and tensors",
                                         "workload": f"configs[4]: {H5}x{W5} low-light frames (no black-level clip, negative DN reach the VST), GuidedResUnet(nf=32), pipeline 'once'"}
        try:
            others["cfg5_fp16_4000x6000"]["roofline"] = leg_roofline(P._plan_of(net5, dev), lambda: P.IterDenoise(f5, net5, arch, pipe),
                                                                     torch.cuda.synchronize, fp16_operands=True)
        except Exception as e:
            others["cfg5_fp16_4000x6000"]["roofline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        del net5, f5
        torch.cuda.empty_cache()
        # SURVEY 8(f) N4 on the same line: one training step at the reference's training shape (runfiles/Gaussian/GRU_5to50_norm_mix.yml:
        # GuidedResUnet nf 32, batch 64 of 256 x 256 Bayer patches), forward + backward + Adam on the HIP kernels (yond_public_amd/train.py)
        try:
            from yond_public_amd.train import TrainStep
            torch.manual_seed(0)
            net6 = getattr(A, arch['name'])(dict(arch, precision='fp32'))
            A.initialize_weights(net6)
            ts6 = TrainStep(net6.to(dev), lr=1e-4, ddp=False)
            g6 = torch.Generator().manual_seed(1)
            hr6 = torch.rand(64, 4, 128, 128, generator=g6).to(dev)
            sg6 = (torch.rand(64, 1, 1, 1, generator=g6) * 0.18 + 0.02).to(dev)
            lr6 = (hr6 + torch.randn(hr6.shape, generator=g6).to(dev) * sg6).clamp(0, 1)
            for _ in range(4):                      # (two eager steps, the hipGraph capture, one replay)
                ts6.step(lr6, hr6, sg6)
            torch.cuda.synchronize()
            t6, n6 = time.perf_counter(), 0
            while n6 < 10 or time.perf_counter() - t6 < 1.0:
                loss6, _ = ts6.step(lr6, hr6, sg6)
                n6 += 1
            torch.cuda.synchronize()
            el6 = time.perf_counter() - t6
            # algorithmic work of a step: forward + data gradient + weight gradient = 3 x the forward's 2 * 201,984 MAC per packed pixel
            # (SURVEY 8d), against the peak of the MFMA the step's dominant kernels issue (fp16, split operands: 3 MFMAs per product)
            fl6 = 3 * 2.0 * 201984 * 64 * 128 * 128
            tf6 = fl6 / (el6 / n6) / 1e12
            others["training_step"] = {"ms_per_step": round(el6 / n6 * 1e3, 3), "patches_per_s": round(n6 * 64 / el6, 1),
                                       "bayer_mp_per_s": round(n6 * 64 * 256 * 256 / 1e6 / el6, 1), "steps": n6, "loss": round(float(loss6), 6),
                                       "loss_scale": getattr(ts6, "last_scale", None),
                                       "roofline": {"bound": "mfma", "achieved": round(tf6, 1), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                    "frac": round(tf6 / PEAK_F16_MFMA_TFLOPS, 4), "algorithmic_tflop_per_step": round(fl6 / 1e12, 3),
                                                    "definition": "3 x forward FLOPs (forward, data gradient, weight gradient) / step time, whole step "
                                                                  "(wall clock, host included)"},
                                       "collectives": {"grad_all_reduce": D.STATS["grad_all_reduce"], "grad_bytes": D.STATS["grad_bytes"],
                                                       "note": "one rank: no gradient exchange; under torchrun two 25 MB buckets per step (distributed.GradReducer)"},
                                       "workload": "SURVEY 8(f) N4: GuidedResUnet(nf=32), batch 64 x [4][128][128] (256 x 256 Bayer patches), L1 loss, Adam; "
                                                   "forward / data gradients and the weight gradients of the 3x3 stride-1 layers on the split-operand "
                                                   "fp16-MFMA kernels (loss-scaled), the other layers on the fp32 MFMA; the step replayed as one hipGraph",
                                       "graph": bool(getattr(ts6, "_graphs", None))}
            del ts6, net6, hr6, lr6, sg6
        except Exception as e:                      # (a reported extra: it must not take the headline line down with it)
            others["training_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    # final metric reduction (the only collective of the eval path): PSNR of the last output vs the clean frame
    dn = res['raw_dns'][-1]
    dn = dn[-1] if dn.dim() == 3 else dn
    mse = torch.mean((dn.double() - last_clean.double()) ** 2).item()
    mse_in = torch.mean((last_noisy.double() - last_clean.double()) ** 2).item()
    sums = D.MetricSums(1)
    sums.update([10 * np.log10(1.0 / mse)], [0.0])
    red = sums.reduce(dev)

    if rank == 0:
        mp = H * W / 1e6
        cfg_idx = {2: 1, 3: 2, 4: 3, 5: 4}[a.cfg]
        if a.batch:
            driver = (f"denoise_stream_batches: per-frame NLE, ONE batched forward of {a.batch} frames, the estimators of batch k+1 on a second HIP stream under the forward of batch k"
                      if (a.mode == "once" and not a.sequential) else f"IterDenoiseBatch: per-frame NLE, ONE batched forward of {a.batch} frames")
        elif a.cfg == 3:
            driver = (f"YOND_SIDD.eval's loop: groups of {a.group} images (round 1 = ONE batch-{32 * a.group} forward, round 2 another; estimates, tables, "
                      "block metrics per image), consecutive groups overlapped on two HIP streams (denoise_stream_groups)" if a.group > 1 else "YOND_SIDD.eval's loop body per image: IterDenoise (batch-32 forwards) + block metrics, one image at a time")
        elif stream_driver and a.mode == "iter":
            driver = (f"denoise_stream: {P.STREAM_LANES} independent in-order lanes (HIP streams), frame k's whole chain (self NLE, pass 1, collaborative NLE, pass 2) on lane k mod "
                      f"{P.STREAM_LANES}: one lane's estimators run under the other's network pass" if (P.STREAM_LANES > 1 and P.LANE_PIPELINES) else
                      "denoise_stream: first and second passes interleaved on the main stream, estimators on a side stream")
        elif stream_driver:
            driver = ("denoise_stream: NLE of frame k+1 on a side HIP stream while the convolutions of frame k run" +
                      (f"; the network passes of consecutive frames alternate between {P.STREAM_LANES} streams (a launch's last, partly filled round of persistent "
                       "workgroups and the gaps between dependent launches are covered by the other frame's launches)" if P.STREAM_LANES > 1 else ""))
        else:
            driver = "IterDenoise, one frame at a time"
        out = {
            "metric": "Bayer megapixels/sec end-to-end (NLE+VST+denoise+iVST)",
            "value": round(world * n_timed * mp / elapsed, 2), "unit": "Bayer MP/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp32": "f32 (3x3 convolutions: f32 operands as two f16 halves -- 22-23 significant bits -- on the f16 MFMA, 3 products per "
                              "f32 product, f32 accumulate; per-layer error vs float64 asserted <= 2x the f32 direct kernel's and <= 2e-5 of max|ref| "
                              "(tests/test_hip_conv.py; measured 1.3e-6 vs 1.5e-6), whole frame vs the reference <= 1e-4 (tests/test_hip_pipeline.py, full_cfg2)",
                      "fp32-mfma": "f32", "fp16": "f16 MFMA operands, f32 accumulate and tensors (cfg 5)"}[a.precision],
            "data": "synthetic",
            "config": {"workload": (f"configs[{cfg_idx}]: SIDD-validation-shaped synthetic items ({SIDD_FULL[0]}x{SIDD_FULL[1]} frame for the round-1 "
                                    f"estimate + 32 blocks of 256x256 Bayer), {F} per step and GPU, each through YOND_SIDD.eval's loop body: "
                                    f"self NLE on the full frame, bias LUT, batch-32 {a.arch}(nf=32) forward, collaborative NLE (SIDD_256), second "
                                    f"pass, per-block PSNR/SSIM of both rounds; `value` counts the {H * W / 1e6:.3f} MP of denoised blocks per image")
                                   if a.cfg == 3 else
                                   f"configs[{cfg_idx}]: {H}x{W} synthetic Poisson-Gaussian Bayer frames"
                                   f"{'' if clip else ' (low light, no black-level clip)'}, {F} per step and GPU, each through the full "
                                   f"NLE+VST+{a.arch}(nf=32)+iVST, pipeline '{a.mode}' ({n_pass} pass(es) per frame), bias_corr=pre, k=29",
                       "frames_per_step_per_gpu": F, "ms_per_frame": round(elapsed / n_timed * 1e3, 3),
                       "timed_region_s": round(elapsed, 3), "warmup_s": round(warm_s, 3),
                       "parallelism": f"image-parallel x{world}", "driver": driver,
                       "weights": {"denoising": "synthetic.denoising_state_dict (analytic 3x3 box-mean path + eps-scaled procedural weights)",
                                   "procedural": "synthetic.procedural_state_dict (seeded random)"}[a.weights]},
            "gfx_clock": sclk_res,
            "hbm_gb": {"peak_allocated": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2), "peak_reserved": round(torch.cuda.max_memory_reserved(dev) / 1e9, 2),
                       "note": "torch's caching allocator on rank 0 over the whole run (every leg of this line); the library allocates nothing itself"},
            "roofline": roof,
            "sequential": seq,
            "iter_pipeline": iter_leg,
            "without_kernel_events": noev,
            "fp32_mfma_path": strict,
            "other_configs": others,
            "roofline_vst_nle": roof_hbm,
            "conv_stack": {"ms_per_frame": round(conv_ms, 3), "tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None,
                           "share_of_frame": round(conv_ms / (elapsed / n_timed * 1e3), 3) if conv_ms else None,
                           "measured": "HIP events around every convolution launch, the median of three instrumented passes after the timed region"},
            "psnr_vs_clean_db": {"denoised": round(red["psnr_last"], 3), "noisy_input": round(10 * np.log10(1.0 / mse_in), 3)},
            "estimated_K_sigma": [[round(float(v), 4) for v in pr] for pr in (res['params'] if not a.batch else res['params'][-1])],
            **({"images_per_s": round(world * n_timed / elapsed, 2), "ms_per_image": round(elapsed / n_timed * 1e3, 3),
                "full_frame_mp_per_s": round(world * n_timed * SIDD_FULL[0] * SIDD_FULL[1] / 1e6 / elapsed, 1),
                "block_metrics_last": {"psnr_iter0": round(float(np.mean(res['metrics'][0][0])), 3), "psnr_iter1": round(float(np.mean(res['metrics'][1][0])), 3),
                                       "ssim_iter0": round(float(np.mean(res['metrics'][0][1])), 4), "ssim_iter1": round(float(np.mean(res['metrics'][1][1])), 4)},
                "reference_s_per_image": {"value": REF_SIDD_S_PER_IMAGE, "source": "logs/log_YOND_SIDD_simple+full_pre_grumix_iter.log:10,130 "
                                          "(204 s / 40 images on the authors' unnamed GPU: context, not a baseline for this hardware)"}} if a.cfg == 3 else {}),
            # torch.distributed traffic of this run (a process group exists whenever the torchrun environment is set,
            # world size 1 included): barrier + max-over-ranks of the timing + the PSNR reduction
            "collectives": dict(D.STATS, world_size=group_world,
                                per_rank_mp_per_s=[round(a.steps * F * H * W / 1e6 / t, 2) for t in per_rank_s],
                                per_rank=[{"rank": r, "mp_per_s": round(a.steps * F * H * W / 1e6 / t, 2), "in_kernel_mhz": round(m, 1) if m else None}
                                          for r, (t, m) in enumerate(zip(per_rank_s, per_rank_mhz))]),
            "static_notes": static_notes(),
        }
        if world == 1 and not a.no_cpu_baseline and not a.batch:
            out["cpu_baseline"], out["parity_vs_oracle"] = cpu_baseline_and_parity(a, arch, dev, make_net)
        print(json.dumps(out), flush=True)
    D.barrier()
    D.finalize()


if __name__ == "__main__":
    main()
