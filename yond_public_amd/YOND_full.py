"""Full-frame evaluation drivers: the entry points README.md:38-47 names (YOND_any / YOND_ELD / YOND_LRID / YOND_DND) but the
reference does not ship.  They are modelled on YOND_SIDD.py with the full-frame runfiles it does ship
(runfiles/YOND/{ANY,ELD,LRID}_simple+full_pre_grumix.yml: `full_dn: True`, `iter`, clip False, `ratio_list`, `cam_list`):

    python YOND_any.py  -f runfiles/YOND/ANY_simple+full_pre_grumix.yml  -m eval
    python YOND_ELD.py  -f runfiles/YOND/ELD_simple+full_pre_grumix.yml  -m eval
    python YOND_LRID.py -f runfiles/YOND/LRID_simple+full_pre_grumix.yml -m eval
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 YOND_ELD.py -f ... -m eval

Per (camera,) ratio of the runfile's list the dataset is switched with `change_eval_ratio` (data_process/yond_datasets.py:912,
1033), every frame goes through `IterDenoise` as ONE whole-frame forward per round (YOND_SIDD.py:387-389, 456-458) with
p = {wp, bl, ratio, scale = (wp - bl) / ratio} (:503-505), and PSNR / SSIM of the whole frame are accumulated per ratio.
Frames are sharded one per GPU process (frame k -> rank k mod world); the metric sums are reduced with ONE all-reduce per
ratio.  Datasets are `.npy`-converted trees (yond_public_amd/data.py: rawpy is not in this image); without any data the
driver runs on seeded synthetic low-light frames of the runfile's H x W so that the control flow can be exercised and timed.
"""
import os
import time

import numpy as np
import torch
import yaml

from . import archs as _archs
from . import data as _data
from . import distributed as D
from . import pipeline as P
from . import synthetic as S
from .YOND_SIDD import YONDParser, log

STREAM_EVAL = True                 # eval(): frames through pipeline.denoise_stream (False: IterDenoise one frame at a time, synchronised after every frame)


class SyntheticFrames:
    """Stand-in when the runfile's root_dir holds no frames: low-light Poisson-Gaussian frames of the runfile's size."""

    def __init__(self, n, H, W, K=2.0, sigma=8.0):
        self.n, self.H, self.W, self.K, self.sigma, self.ratio = n, H, W, K, sigma, 1

    def __len__(self):
        return self.n

    def change_eval_ratio(self, *a, **kw):
        self.ratio = kw.get('ratio', a[-1] if a else 1)

    def __getitem__(self, k):
        rng = np.random.default_rng(4000 + k)
        clean = (S.synth_clean(self.H, self.W) * (0.6 / self.ratio)).astype(np.float32)       # exposure 1 / ratio ...
        noisy = (rng.poisson(clean * 959.0 / self.K) * self.K + rng.normal(0.0, self.sigma, clean.shape)) / 959.0
        return {'lr': (noisy * self.ratio).astype(np.float32), 'hr': (clean * self.ratio).astype(np.float32),   # ... digital gain = ratio
                'name': f'synthetic_{k:03d}_x{self.ratio:02d}', 'ratio': self.ratio, 'cfa': 'rggb', 'meta': None}


class YOND_Full:
    def __init__(self, args=None):
        self.parser = YONDParser().parse(args)
        with open(self.parser.runfile, 'r', encoding='utf-8') as f:
            self.args = yaml.load(f.read(), Loader=yaml.FullLoader)
        self.mode = self.args['mode'] if self.parser.mode is None else self.parser.mode
        self.rank, self.local_rank, self.world = D.init()
        if not torch.cuda.is_available():
            raise SystemExit("the full-frame drivers need an MI355X: the HIP path has no CPU fallback")
        self.device = torch.device('cuda', self.local_rank)
        torch.cuda.set_device(self.device)
        self.arch, self.pipe = self.args['arch'], dict(self.args['pipeline'])
        if self.pipe.get('bias_corr') == 'none':
            self.pipe['bias_corr'] = None
        self.pipe.setdefault('k', 29)                                   # (the ANY runfile leaves k to the driver's default)
        if not self.pipe.get('full_dn', False):
            raise SystemExit(f"{self.parser.runfile}: the full-frame drivers run runfiles with `full_dn: True` (YOND_SIDD.py handles the block layout)")
        self.model_name, self.method_name = self.args['model_name'], self.args['method_name']
        os.makedirs('./logs', exist_ok=True)
        self.logfile = f'./logs/log_{self.method_name}.log' if self.rank == 0 else None
        self.net = getattr(_archs, self.arch['name'])(self.arch)
        for suffix in ('_best_model.pth', '_last_model.pth', '.pth'):                         # YOND_SIDD.py:178-182
            model_path = f"{self.args['fast_ckpt']}/{self.model_name}{suffix}"
            if os.path.exists(model_path):
                self.net.load_state_dict(torch.load(model_path, map_location='cpu'))
                break
        else:
            model_path = None
            self.net.load_state_dict(S.denoising_state_dict(self.net, 0))
        self.net = self.net.to(self.device).eval()
        self.biaslut = P.BiasLUT() if os.path.exists('checkpoints/bias_lut_2d.npy') else None
        if self.rank == 0:
            log(f'Method Name:\t{self.method_name}', self.logfile, notime=True)
            log(f'Checkpoint:\t{model_path or "none found -> synthetic denoising weights (timing / parity only)"}', self.logfile, notime=True)
            log(f"Let's use {self.world} GPUs (one process each, image-parallel)!", self.logfile, notime=True)
        self.change_eval_dst('test' if 'test' in self.mode else 'eval')

    def change_eval_dst(self, mode='eval'):
        self.dst = dict(self.args[f'dst_{mode}'])
        cls = self.dst.get('dataset', 'Any_Dataset')
        table = {'ELD_Full_Dataset': _data.ELD_Full_Dataset, 'LRID_Dataset': _data.LRID_Dataset, 'Any_Dataset': _data.Any_Dataset}
        self.dst_eval = None
        if cls in table and os.path.isdir(str(self.dst.get('root_dir', ''))):
            ds = table[cls](self.dst)
            if len(ds) or cls != 'Any_Dataset':
                self.dst_eval = ds
        if self.dst_eval is None or (len(self.dst_eval) == 0 and cls == 'Any_Dataset'):
            self.dst_eval = SyntheticFrames(self.parser.synthetic, int(self.dst.get('H', 3472)), int(self.dst.get('W', 4624)))

    def IterDenoise(self, data, params):
        return P.IterDenoise(data['lr'], self.net, self.arch, self.pipe, p=params['p'], device=self.device,
                             log=(lambda s: log(s, self.logfile)) if self.parser.verbose else None, biaslut=self.biaslut)

    def _sweeps(self):
        """(label, switch) per evaluated subset: ratio_list x cam_list as the runfile gives them."""
        ratios = self.dst.get('ratio_list', [self.dst.get('ratio', 1)])
        cams = self.dst.get('cam_list', [None])
        for cam in cams:
            for ratio in ratios:
                if cam is None:
                    yield f'x{ratio}', (lambda r=ratio: self.dst_eval.change_eval_ratio(ratio=r))
                else:
                    yield f'{cam} x{ratio}', (lambda c=cam, r=ratio: self.dst_eval.change_eval_ratio(c, ratio=r)
                                              if not isinstance(self.dst_eval, SyntheticFrames) else self.dst_eval.change_eval_ratio(ratio=r))

    def eval(self, epoch=-1):
        n_it = self.pipe['max_iter'] + 1 if self.pipe.get('iter') == 'iter' else 1
        results = {}
        self.metrics = {}
        for label, switch in self._sweeps():
            switch()
            ds = self.dst_eval
            wp, bl = float(getattr(ds, 'wp', self.dst.get('wp', 1023))), float(getattr(ds, 'bl', self.dst.get('bl', 64)))
            sums = D.MetricSums(n_it)
            mine = D.shard_dataset(ds, self.rank, self.world)
            torch.cuda.synchronize()
            t0, t_path, npix = time.perf_counter(), 0.0, 0
            from .data import Prefetcher                     # loader threads read / convert / upload the frames ahead of the GPU
            loader = Prefetcher(ds, mine, self.device, upload=('lr', 'hr'), depth=getattr(self.parser, 'prefetch', 4), workers=getattr(self.parser, 'loaders', 4))

            def params_of(data):
                p = dict(self.pipe)
                p.update({'wp': wp, 'bl': bl, 'ratio': data.get('ratio', 1), 'gain': 1, 'sigma': 0})           # YOND_SIDD.py:503-505
                p['scale'] = (p['wp'] - p['bl']) / p['ratio']
                return p

            def account(k, data, res):
                psnrs, ssims = [], []
                if data.get('hr') is not None:
                    hr = data['hr'] if isinstance(data['hr'], torch.Tensor) else torch.from_numpy(np.ascontiguousarray(data['hr'], np.float32)).to(self.device)
                    H, W = hr.shape
                    for dn in res['raw_dns']:
                        ps, ss = P.block_metrics(dn, hr.clamp(0, 1), bh=H, bw=W)          # whole-frame PSNR / SSIM (data range 1)
                        psnrs.append(float(np.mean(ps)))
                        ssims.append(float(np.mean(ss)))
                    sums.update(psnrs, ssims)
                self.metrics[data['name']] = {'psnr': psnrs, 'ssim': ssims, 'reg': res['regs']}
                log(f"[rank {self.rank}] {data['name']}: " + (f"PSNR={psnrs[-1]:.2f}, SSIM={ssims[-1]:.4f}" if psnrs else "denoised (no reference frame)")
                    + f", K={res['params'][-1][0]:.3f}, sigma={res['params'][-1][1]:.3f}", self.logfile)

            streamed = STREAM_EVAL and getattr(self.parser, 'stream', True) and not self.parser.verbose and P.stream_applies(self.pipe, self.pipe, self.biaslut)
            if streamed:
                # pipeline.denoise_stream: the estimator of the next frame on a side stream, the network passes of consecutive frames on two lanes -- every
                # frame's result is IterDenoise's (tests/test_hip_pipeline.py); results arrive a few frames late, in order
                queue = []

                def feed():
                    for k, data in loader:
                        queue.append((k, data))
                        yield data['lr'], params_of(data)
                t1 = time.perf_counter()
                for res in P.denoise_stream(feed(), self.net, self.arch, self.pipe, device=self.device):
                    k, data = queue.pop(0)
                    account(k, data, res)
                    npix += int(np.prod(tuple(data['lr'].shape)))
                torch.cuda.synchronize()
                t_path += time.perf_counter() - t1           # (includes what the loop waited for its loader threads)
            else:
                for k, data in loader:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    res = self.IterDenoise(data, {'p': params_of(data), 'img_id': k})
                    account(k, data, res)
                    torch.cuda.synchronize()
                    t_path += time.perf_counter() - t1
                    npix += int(np.prod(tuple(data['lr'].shape)))
            torch.cuda.synchronize()
            dt = D.max_over_ranks(time.perf_counter() - t0, self.device)
            red = sums.reduce(self.device)
            results[label] = red
            if self.rank == 0:
                log(f'{self.method_name} [{label}]: {len(ds)} frames', self.logfile)
                if red['count']:
                    for it in range(n_it):
                        log(f"Iter{it}: PSNR={red[f'psnr_iter{it}']:.2f}, SSIM={red[f'ssim_iter{it}']:.4f}", self.logfile)
                    log(f"Iter_last: PSNR={red['psnr_last']:.2f}, SSIM={red['ssim_last']:.4f}", self.logfile)
                log(f"{len(ds)} frames on {self.world} GPU(s) in {dt:.2f} s (rank 0: {t_path / max(len(mine), 1) * 1e3:.1f} ms per frame in "
                    + ("denoise_stream + metrics, waits for the loader threads included" if streamed else "IterDenoise + metrics")
                    + f" = {npix / 1e6 / max(t_path, 1e-9):.0f} Bayer MP/s" + ("" if streamed else "; the rest is data loading") + ")", self.logfile)
        if self.rank == 0:
            log(f"collectives: backend={D.STATS['backend']}, all_reduce={D.STATS['all_reduce']}, barrier={D.STATS['barrier']}", self.logfile)
        return results


def main(argv=None):
    trainer = YOND_Full(argv)
    try:
        return trainer.eval(-1)
    finally:
        D.finalize()


if __name__ == '__main__':
    main()
