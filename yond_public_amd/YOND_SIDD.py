"""Evaluation entry point with the reference's surface (YOND_SIDD.py:136-236, 485-570, 723-744):

    python YOND_SIDD.py -f runfiles/YOND/SIDD_simple+full_pre_grumix.yml -m eval
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 YOND_SIDD.py -f ... -m eval

Same runfile schema (`pipeline`, `dst*`, `arch`), same class / method names (`YOND_SIDD.eval`, `IterDenoise`,
`VST_Denoiser`, `Simple_Denoiser`), same model lookup by name and checkpoint search order.  Differences that
are the point of this build: every per-pixel pass runs on the MI355X HIP kernels; images are sharded one per
GPU process (image k -> rank k mod world) instead of nn.DataParallel over batch 1; per-block PSNR/SSIM are
computed on the device and reduced with ONE all-reduce at the end.

Datasets: the reference's SIDD layout (`<root_dir>/SIDD_Validation_Raw/Validation{Noisy,Gt}BlocksRaw.mat`, MATLAB v5, plus
`SIDD_Benchmark_Data/*/*_010.MAT` metadata: yond_public_amd/data.py mirrors data_process/yond_datasets.py:767-868), or
`<root_dir>/npy/{noisy,gt}_{k:03d}.npy` ((32,256,256) float32 in [0,1], plus optional `full_{k:03d}.npy`); DNG / ARW and
MATLAB v7.3 need rawpy / h5py, which this image lacks.  Without data the driver evaluates seeded synthetic stand-ins so
the whole control flow can be exercised and timed.
"""
import argparse
import os
import time
from pathlib import Path

import numpy as np
import torch
import yaml

from . import archs as _archs
from . import distributed as D
from . import pipeline as P
from . import synthetic as S


def log(string, log=None, notime=False):
    s = f'{time.strftime("%Y-%m-%d %H:%M:%S")} >>  {string}' if not notime else string
    print(s, flush=True)
    if log is not None:
        with open(log, 'a+') as f:
            f.write(s + '\n')


class SyntheticSIDD:
    """Stand-in for SIDD_Dataset (data_process/yond_datasets.py:767-868): items with 'lr', 'hr' of shape
    (32, 256, 256), 'lr_full' (the frame used for the round-1 estimate: 3000 x 5328, the size SURVEY 8d names for the
    stand-ins of the SIDD full frames), 'name'."""

    def __init__(self, n=40, K=4.0, sigma=6.0, full_hw=(3000, 5328), distinct=8):
        """distinct: item k is the synthesis of seed k % distinct, kept once made (a 16 MP Poisson draw takes ~1 s of NumPy -- 300 x the
        GPU time of the image; a real dataset's items come from files, which the loader threads overlap with the GPU)."""
        self.n, self.K, self.sigma, self.full_hw, self.distinct = n, K, sigma, full_hw, max(1, distinct)
        self._made = {}

    def prepare(self, indices=None, workers=8):
        """Synthesise the distinct items up front, in parallel -- the counterpart of SIDD_Dataset.__init__ reading the validation blocks
        before the evaluation loop starts (data_process/yond_datasets.py:797-806); afterwards an item costs no host arithmetic."""
        from concurrent.futures import ThreadPoolExecutor
        need = sorted({k % self.distinct for k in (indices if indices is not None else range(self.n))} - set(self._made))
        if need:
            with ThreadPoolExecutor(max_workers=min(workers, len(need))) as ex:
                list(ex.map(self.__getitem__, need))

    def __len__(self):
        return self.n

    def item_size(self, k):
        return self.full_hw[0] * self.full_hw[1]

    def __getitem__(self, k):
        j = k % self.distinct
        if j not in self._made:
            noisy, clean = S.synth_noisy(256, 8192, self.K, self.sigma, 100 + j)
            full, _ = S.synth_noisy(self.full_hw[0], self.full_hw[1], self.K, self.sigma, 500 + j)
            self._made[j] = (np.array(np.split(noisy, 32, axis=-1)), np.array(np.split(clean, 32, axis=-1)), full)
        lr, hr, full = self._made[j]
        return {'lr': lr, 'hr': hr, 'lr_full': full, 'name': f'synthetic_{k:03d}', 'meta': None, 'cfa': 'rggb'}


class NpySIDD:
    def __init__(self, root):
        self.root = root
        self.ids = sorted(int(p.stem.split('_')[1]) for p in Path(root).glob('noisy_*.npy'))

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        k = self.ids[i]
        d = {'lr': np.load(f'{self.root}/noisy_{k:03d}.npy').astype(np.float32), 'name': f'sidd_{k:03d}', 'meta': None, 'cfa': 'rggb'}
        if os.path.exists(f'{self.root}/gt_{k:03d}.npy'):
            d['hr'] = np.load(f'{self.root}/gt_{k:03d}.npy').astype(np.float32)
        d['lr_full'] = np.load(f'{self.root}/full_{k:03d}.npy').astype(np.float32) if os.path.exists(f'{self.root}/full_{k:03d}.npy') else None
        return d


class YOND_SIDD:
    def __init__(self, args=None):
        self.parser = YONDParser().parse(args)
        self.initialization()

    def initialization(self):
        with open(self.parser.runfile, 'r', encoding='utf-8') as f:
            self.args = yaml.load(f.read(), Loader=yaml.FullLoader)
        self.mode = self.args['mode'] if self.parser.mode is None else self.parser.mode
        self.rank, self.local_rank, self.world = D.init()
        if not torch.cuda.is_available():
            raise SystemExit("YOND_SIDD needs an MI355X: the HIP path has no CPU fallback")
        self.device = torch.device('cuda', self.local_rank)
        torch.cuda.set_device(self.device)
        self.dst, self.arch, self.pipe = self.args['dst'], self.args['arch'], self.args['pipeline']
        if self.pipe['bias_corr'] == 'none':
            self.pipe['bias_corr'] = None
        self.model_name, self.method_name = self.args['model_name'], self.args['method_name']
        self.fast_ckpt = self.args['fast_ckpt']
        os.makedirs('./logs', exist_ok=True)
        self.logfile = f'./logs/log_{self.method_name}.log' if self.rank == 0 else None
        # model: looked up by name, checkpoint search order best -> last -> plain (YOND_SIDD.py:177-184)
        self.net = getattr(_archs, self.arch['name'])(self.arch)
        for suffix in ('_best_model.pth', '_last_model.pth', '.pth'):
            model_path = f'{self.fast_ckpt}/{self.model_name}{suffix}'
            if os.path.exists(model_path):
                state = torch.load(model_path, map_location='cpu')
                self.net.load_state_dict(state)
                break
        else:
            model_path = None                      # no checkpoint: a deterministic weak denoiser (both rounds run)
            self.net.load_state_dict(S.denoising_state_dict(self.net, 0))
        self.net = self.net.to(self.device).eval()
        # the 2-D bias LUT when its blob is present (YOND_SIDD.py:171), else get_bias per image
        self.biaslut = P.BiasLUT() if os.path.exists('checkpoints/bias_lut_2d.npy') else None
        if self.rank == 0:
            nparam = sum(p.numel() for p in self.net.parameters())
            log(f'Method Name:\t{self.method_name}', self.logfile, notime=True)
            log(f'Architecture:\t{self.arch["name"]}', self.logfile, notime=True)
            log(f'Parameters:\t{nparam / 1e6:.2f}M', self.logfile, notime=True)
            log(f'Checkpoint:\t{model_path or "none found -> synthetic denoising weights (timing / parity only)"}', self.logfile, notime=True)
            log(f"Let's use {self.world} GPUs (one process each, image-parallel)!", self.logfile, notime=True)
        self.change_eval_dst('eval')

    def change_eval_dst(self, mode='eval'):
        self.dst = self.args[f'dst_{mode}']
        root = os.path.join(self.dst['root_dir'], 'npy')
        mat = os.path.join(self.dst['root_dir'], 'SIDD_Validation_Raw',
                           'ValidationNoisyBlocksRaw.mat' if mode == 'eval' else 'BenchmarkNoisyBlocksRaw.mat')
        if os.path.exists(mat):                       # the reference's layout (data_process/yond_datasets.py:797-806)
            from .data import SIDD_Dataset
            self.dst_eval = SIDD_Dataset(dict(self.dst, mode=mode))
        elif os.path.isdir(root) and list(Path(root).glob('noisy_*.npy')):
            self.dst_eval = NpySIDD(root)
        else:
            self.dst_eval = SyntheticSIDD(self.parser.synthetic)

    # -- reference method names -----------------------------------------------------------------
    def Simple_Denoiser(self, lr_raw):
        return P.Simple_Denoiser(lr_raw, self.net, device=self.device)

    def VST_Denoiser(self, lr_raw, hr_raw=None, bias_corr='pre', bias_func=None, denoiser='gru32n', p=None):
        return P.VST_Denoiser(lr_raw, p, self.net, self.arch, bias_corr, bias_func, self.pipe.get('vst_type', 'exact'),
                              device=self.device, biaslut=self.biaslut)

    def IterDenoise(self, data, params):
        # data['lr'] is the [32][256][256] stack; pipeline.IterDenoise concatenates it for the estimate (:315) and, with
        # pipe['full_dn'], for the denoiser (:388); the collaborative estimate re-tiles per block (SIDD_256, :431)
        res = P.IterDenoise(data['lr'], self.net, self.arch, self.pipe, lr_full=data.get('lr_full'), p=params['p'],
                            device=self.device, log=(lambda s: log(s, self.logfile)) if self.parser.verbose else None,
                            biaslut=self.biaslut,
                            est={'root_dir': (getattr(self, 'dst', None) or {}).get('root_dir'), 'img_id': params.get('img_id'),
                                 'name': data.get('name')})
        cat = lambda a: torch.cat(list(a), dim=-1) if isinstance(a, torch.Tensor) else np.concatenate(a, axis=-1)   # (:480-481)
        res['lr_raw'] = cat(data['lr'])
        res['hr_raw'] = cat(data['hr']) if data.get('hr') is not None else None
        return res

    def IterDenoiseGroup(self, datas, params_list):
        """IterDenoise for several items at once (pipeline.IterDenoiseGroup): round 1 of the G images as ONE batch-(32 G) forward, round 2
        likewise; every image keeps its own estimates, tables and t, and its result is IterDenoise's.  The reference takes the
        images one by one (:507-514); they are independent, and 32 blocks of 128 x 128 packed pixels leave the deep levels of a forward
        (16 x 16 and 8 x 8 pixels) far too small for the chip."""
        ress = P.IterDenoiseGroup([(d['lr'], d.get('lr_full')) for d in datas], self.net, self.arch, self.pipe, ps=[q['p'] for q in params_list],
                                  device=self.device, log=(lambda s: log(s, self.logfile)) if self.parser.verbose else None, biaslut=self.biaslut,
                                  ests=[{'root_dir': (getattr(self, 'dst', None) or {}).get('root_dir'), 'img_id': q.get('img_id'), 'name': d.get('name')}
                                        for d, q in zip(datas, params_list)])
        cat = lambda a: torch.cat(list(a), dim=-1) if isinstance(a, torch.Tensor) else np.concatenate(a, axis=-1)   # (:480-481)
        for res, d in zip(ress, datas):
            res['lr_raw'] = cat(d['lr'])
            res['hr_raw'] = cat(d['hr']) if d.get('hr') is not None else None
        return ress

    def _groups(self, it):
        """The prefetcher's items `group` at a time (consecutive items; the last group may be short)."""
        G = max(1, int(getattr(self.parser, 'group', 1)))
        batch = []
        for k, data in it:
            batch.append((k, data))
            if len(batch) == G:
                yield batch
                batch = []
        if batch:
            yield batch

    def eval(self, epoch=-1):
        n_it = self.pipe['max_iter'] + 1 if self.pipe.get('iter') == 'iter' else 1
        sums = D.MetricSums(n_it)
        p = dict(self.pipe)
        p.update({'wp': 1023, 'bl': 64, 'ratio': 1, 'gain': 1, 'sigma': 0})          # YOND_SIDD.py:504
        p['scale'] = (p['wp'] - p['bl']) / p['ratio']
        mine = D.shard_dataset(self.dst_eval, self.rank, self.world)        # size-aware (five phones, five frame sizes) where sizes are known
        self.metrics = {}
        if hasattr(self.dst_eval, 'prepare'):
            self.dst_eval.prepare(mine)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_path = 0.0
        marks = []                                          # (wall clock, path seconds so far) behind every image: the steady state is the second half
        # the items are read and uploaded by loader threads ahead of the GPU (the reference reads each in front of its IterDenoise,
        # :507-514): file reads, the float32 conversion and the 64 MB upload of the full frame overlap the previous images' kernels
        from .data import Prefetcher
        G = max(1, int(getattr(self.parser, 'group', 1)))
        pf = Prefetcher(self.dst_eval, mine, self.device, depth=max(self.parser.prefetch, 3 * G), workers=self.parser.loaders)
        est_type = str(self.pipe.get('est_type', 'simple'))
        streamed = (P.STREAM_GROUPS and getattr(self.parser, 'stream', True) and self.pipe.get('iter') == 'iter' and self.pipe.get('max_iter', 1) == 1
                    and 'simple' in est_type and 'cal_est' not in self.pipe and 'rot_cfa' not in p and self.biaslut is None)

        def metrics_of(data, res):
            psnrs, ssims = [], []
            hr_raw = data.get('hr')
            if hr_raw is not None:
                hr = torch.cat(list(hr_raw), dim=-1) if isinstance(hr_raw, torch.Tensor) else torch.from_numpy(np.concatenate(hr_raw, axis=-1)).to(self.device)
                for dn in res['raw_dns']:
                    ps, ss = P.block_metrics(dn, hr)                                   # :649-652 per 256x256 block
                    psnrs.append(float(np.mean(ps)))
                    ssims.append(float(np.mean(ss)))
            return psnrs, ssims

        def account(data, res, psnrs, ssims):
            if psnrs:
                sums.update(psnrs, ssims)            # iterations that did not run count -1 in their own meter (:644-647)
            self.metrics[data['name']] = {'psnr': psnrs, 'ssim': ssims, 'reg': res['regs']}
            log(f"[rank {self.rank}] {data['name']}: PSNR={psnrs[-1] if psnrs else float('nan'):.2f}, "
                f"SSIM={ssims[-1] if ssims else float('nan'):.4f}", self.logfile)

        if streamed:
            # consecutive groups overlapped on two HIP streams (pipeline.denoise_stream_groups): group k+1's full-frame estimates under group k's first
            # pass, group k's collaborative estimates under group k-1's second; the block metrics of a finished group on a third stream
            batches, done = {}, {}

            def groups():
                for gi, batch in enumerate(self._groups(pf)):
                    batches[gi] = batch
                    yield [(d['lr'], d.get('lr_full')) for _, d in batch]

            def finish(gi, ress):
                done[gi] = [metrics_of(d, r) for (_, d), r in zip(batches[gi], ress)]

            for gi, ress in enumerate(P.denoise_stream_groups(groups(), self.net, self.arch, self.pipe, p=dict(p, cfa=[[1, 2], [2, 3]]), device=self.device,
                                                             log=(lambda s: log(s, self.logfile)) if self.parser.verbose else None, finish=finish)):
                batch = batches.pop(gi)
                for (k, data), res, (psnrs, ssims) in zip(batch, ress, done.pop(gi)):
                    account(data, res, psnrs, ssims)
                t_path = time.perf_counter() - t0          # (groups overlap: the path IS the wall clock here)
                marks.extend([(time.perf_counter(), t_path)] * len(batch))
        else:
          for batch in self._groups(pf):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            datas = [d for _, d in batch]
            plist = [{'p': dict(p, cfa=d.get('cfa', [[1, 2], [2, 3]])), 'img_id': k} for k, d in batch]      # YOND_SIDD.py:510
            if len(batch) == 1:
                ress = [self.IterDenoise(datas[0], plist[0])]
            else:
                ress = self.IterDenoiseGroup(datas, plist)
            for (k, data), res in zip(batch, ress):
                psnrs, ssims = metrics_of(data, res)
                account(data, res, psnrs, ssims)
            torch.cuda.synchronize()
            t_path += time.perf_counter() - t1              # estimate + denoise (+ metrics) of this group's images
            marks.extend([(time.perf_counter(), t_path)] * len(batch))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        red = sums.reduce(self.device)                 # the ONE collective of the eval path (RCCL over xGMI)
        dt = D.max_over_ranks(dt, self.device)
        if self.rank == 0:
            log(f'{self.method_name}:', self.logfile)
            for it in range(n_it):
                log(f"Iter{it}: PSNR={red[f'psnr_iter{it}']:.2f}, SSIM={red[f'ssim_iter{it}']:.4f}", self.logfile)
            log(f"Iter_last: PSNR={red['psnr_last']:.2f}, SSIM={red['ssim_last']:.4f}", self.logfile)
            log(f"{red['count']} images on {self.world} GPU(s) in {dt:.2f} s "
                f"(rank 0: {dt / max(len(mine), 1) * 1e3:.1f} ms wall per image, {t_path / max(len(mine), 1) * 1e3:.1f} ms of it in IterDenoise + metrics; "
                f"the rest is waiting for the {self.parser.loaders} loader threads)", self.logfile)
            self.last_timing = {'wall_ms_per_image': dt / max(len(mine), 1) * 1e3, 'path_ms_per_image': t_path / max(len(mine), 1) * 1e3}
            if len(marks) >= 8:
                h = len(marks) // 2                           # second half: lazily built plans / buffers and the loaders' start-up are behind us
                wall2 = (marks[-1][0] - marks[h - 1][0]) / (len(marks) - h) * 1e3
                path2 = (marks[-1][1] - marks[h - 1][1]) / (len(marks) - h) * 1e3
                self.last_timing.update(steady_wall_ms_per_image=wall2, steady_path_ms_per_image=path2)
                log(f"steady state (second half of rank 0's images): {wall2:.2f} ms wall per image, {path2:.2f} ms of it in IterDenoise + metrics "
                    f"(x{wall2 / max(path2, 1e-9):.2f})", self.logfile)
            log(f"collectives: backend={D.STATS['backend']}, all_reduce={D.STATS['all_reduce']}, barrier={D.STATS['barrier']}", self.logfile)
        return red


    def benchmark(self):
        """YOND_SIDD.py:572-630 (`-m test`): the SIDD BENCHMARK blocks (no ground truth) through IterDenoise; keeps what the
        reference keeps -- per-image estimates as `reg_test` in self.metrics and the two submission arrays
        bench_init / bench_results [N][32][256][256] (first / last round) -- and writes them to npy/<method>/ (the reference's
        .mat export is commented out upstream, :615-621; sRGB previews need cv2 and are out of scope).  Images are sharded over
        the ranks; every rank writes the blocks of its own images."""
        n = len(self.dst_eval)
        bench_init = np.zeros((n, 32, 256, 256), np.float32)
        bench_results = np.zeros((n, 32, 256, 256), np.float32)
        p = dict(self.pipe)
        p.update({'wp': 1023, 'bl': 64, 'ratio': 1, 'gain': 1, 'sigma': 0})                  # :586
        p['scale'] = (p['wp'] - p['bl']) / p['ratio']
        self.metrics = getattr(self, 'metrics', None) or {}
        mine = D.shard_dataset(self.dst_eval, self.rank, self.world)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        from .data import Prefetcher
        G = max(1, int(getattr(self.parser, 'group', 1)))
        for batch in self._groups(Prefetcher(self.dst_eval, mine, self.device, depth=max(self.parser.prefetch, 2 * G), workers=self.parser.loaders)):
            datas = [d for _, d in batch]
            plist = [{'p': dict(p, cfa=d.get('cfa', [[1, 2], [2, 3]])), 'img_id': k} for k, d in batch]
            ress = [self.IterDenoise(datas[0], plist[0])] if len(batch) == 1 else self.IterDenoiseGroup(datas, plist)
            for (k, data), res in zip(batch, ress):
                self.metrics.setdefault(data['name'], {})['reg_test'] = res['regs']                   # :597
                first, last = res['raw_dns'][0].cpu().numpy(), res['raw_dns'][-1].cpu().numpy()
                bench_init[k] = np.array(np.split(first, 32, axis=-1))                                # :612-613
                bench_results[k] = np.array(np.split(last, 32, axis=-1))
                log(f"[rank {self.rank}] {data['name']}: {len(res['raw_dns'])} round(s), regs {[tuple(float(v) for v in r) for r in res['regs']]}", self.logfile)
        torch.cuda.synchronize()
        dt = D.max_over_ranks(time.perf_counter() - t0, self.device)
        os.makedirs(f'npy/{self.method_name}', exist_ok=True)
        tag = '' if self.world == 1 else f'_rank{self.rank}'
        np.save(f'npy/{self.method_name}/benchmark_init{tag}.npy', bench_init)
        np.save(f'npy/{self.method_name}/benchmark_results{tag}.npy', bench_results)
        D.barrier()
        if self.rank == 0:
            log(f'{self.method_name} benchmark: {n} images on {self.world} GPU(s) in {dt:.2f} s -> npy/{self.method_name}/benchmark_*.npy', self.logfile)
        return bench_init, bench_results


class YONDParser:
    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)

    def parse(self, args=None):
        a = self.parser
        a.add_argument('--runfile', '-f', default="runfiles/YOND/SIDD_simple+full_pre_grumix.yml", type=Path, help="path to config")
        a.add_argument('--mode', '-m', default='eval', type=str, help="eval or test")
        a.add_argument('--debug', action='store_true', default=False)
        a.add_argument('--nofig', action='store_true', default=True, help="don't save plots (no sRGB rendering in this build)")
        a.add_argument('--nohost', action='store_true', default=False)
        a.add_argument('--gpu', default="0", help="kept for CLI compatibility; ranks pick their device from LOCAL_RANK")
        a.add_argument('--synthetic', type=int, default=40, help="number of synthetic stand-in images when no dataset is found "
                       "(40 = the SIDD validation set's size; each with a 3000 x 5328 frame for the round-1 estimate)")
        a.add_argument('--verbose', action='store_true', default=False)
        a.add_argument('--loaders', type=int, default=4, help="loader threads that read and upload the items ahead of the GPU")
        a.add_argument('--prefetch', type=int, default=4, help="items the loader threads may be ahead of the GPU (at least two groups)")
        a.add_argument('--no-stream', dest='stream', action='store_false', default=True, help="one group at a time instead of consecutive groups overlapped on two HIP streams")
        a.add_argument('--group', type=int, default=4, help="images denoised together: round 1 of a group is ONE batch-(32 x group) forward, round 2 another "
                       "(per image the results are those of --group 1)")
        return a.parse_args(args)


def main(argv=None):
    trainer = YOND_SIDD(argv)
    try:
        out = None
        if 'eval' in trainer.mode:                           # YOND_SIDD.py:736-744
            trainer.change_eval_dst('eval')
            out = trainer.eval(-1)
        if 'test' in trainer.mode:
            trainer.change_eval_dst('test')
            out = trainer.benchmark()
        return out
    finally:
        D.finalize()


if __name__ == '__main__':
    main()
