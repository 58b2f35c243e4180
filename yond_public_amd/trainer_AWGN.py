"""Training entry point with the reference's surface (trainer_AWGN.py:12-76, 78-193, 195-310, 366-404; trainer_base.py:29-135):

    python trainer_AWGN.py -f runfiles/Gaussian/GRU_5to50_norm_mix.yml -m train
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 trainer_AWGN.py -f ... -m train

Same runfile schema (`dst*`, `arch`, `hyper`), same class / method names (`AWGN_Trainer.train / eval / preprocess /
change_eval_dst`), same checkpoint names and search order, same epoch loop: per batch `preprocess` -> forward -> loss ->
backward -> Adam step -> running train PSNR; per epoch scheduler.step(), every `save_freq` epochs the state_dict, every
`plot_freq` epochs a fast eval at sigma_list[1] with the best model kept.  What differs is the point of this build: forward,
backward, loss and Adam run on the HIP kernels (yond_public_amd/train.py), one process per GPU with the gradients averaged
over RCCL in 25 MB buckets behind the backward pass (distributed.GradReducer) instead of torch DDP, every rank draws its own
shard of the epoch's permutation (DistributedSampler's rule), and the evaluation forward is the inference engine.

Data: the reference's training sets are sRGB crops pushed through `unprocess` + `mosaic` on the host (data_process/, outside
SURVEY section 8).  Here a dataset directory holds patches that are ALREADY packed raw: `<root_dir>/<mode>[_<subname>]/*.npy`,
each (4, h, w) or (h, w, 4) float in [0, 1]; the noise model on top is the reference's (yond_datasets.py:308-320:
sigma log-uniform in [sigma_min, sigma_max] / 255 when training, the fixed `dst_eval.sigma` with a per-index seed otherwise;
Bayer-pattern augmentation by rotating the mosaic, :296-303).  Without data the trainer draws seeded synthetic patches so that
the loop can be exercised and timed.  Plots, FastISP previews and the consistency branch (`command: consistency`) are not built.
"""
import argparse
import math
import os
import pickle as pkl
import time
from pathlib import Path

import numpy as np
import torch
import yaml

from . import archs as _archs
from . import distributed as D
from . import synthetic as S
from . import train as T
from .YOND_SIDD import log


def _pack(bayer):
    """(H, W) -> (H/2, W/2, 4), channel = 2 * dy + dx: host-side data preparation (the reference's loader workers)."""
    return np.stack([bayer[0::2, 0::2], bayer[0::2, 1::2], bayer[1::2, 0::2], bayer[1::2, 1::2]], axis=-1)


def bayer_aug(rggb, k=0):
    """data_process/yond_datasets.py:15-19 on an (h, w, 4) packed patch: rotate the mosaic by k quarter turns."""
    h, w, _ = rggb.shape
    bayer = np.empty((2 * h, 2 * w), dtype=rggb.dtype)
    for c in range(4):
        bayer[c // 2::2, c % 2::2] = rggb[..., c]
    return _pack(np.rot90(bayer, k=k, axes=(-2, -1)))


class AverageMeter:
    """utils/utils.py:100-126 without the plots."""

    def __init__(self, name, last_epoch=0):
        self.name, self.history, self.last_epoch = name, [], last_epoch
        self.val = self.avg = self.sum = self.count = 0

    def reset(self):
        if self.avg > 0:
            self.history.append(self.avg)
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class RGB_Img2Raw_Dataset:
    """Item layout of data_process/yond_datasets.py:215-327 ({'name', 'lr', 'hr', 'sigma', 'pattern'}; lr / hr are (4, h, w))
    over packed-raw patches (module docstring).  `self.sigma` is set by the trainer before an evaluation pass (:188, :399)."""

    def __init__(self, args, synthetic=0):
        self.args = dict(args)
        self.mode = self.args['mode']
        d = f"{self.args['root_dir']}/{self.mode}"
        if self.mode == 'train':
            d += f"_{self.args.get('subname', '')}" if 'subname' in self.args else ''
        self.datapath = sorted(str(p) for p in Path(d).glob('*.npy')) if os.path.isdir(d) else []
        self.names = [os.path.basename(p)[:-4] for p in self.datapath]
        self.synthetic = 0 if self.datapath else int(synthetic)
        if not self.datapath and not self.synthetic:
            raise FileNotFoundError(f"no packed-raw patches (*.npy) under {d}; pass --synthetic N to train on synthetic patches")
        self.buffer = [None] * len(self)
        self.h, self.w = self.args['H'] // 2, self.args['W'] // 2
        self.sigma = -1

    def __len__(self):
        return len(self.datapath) or self.synthetic

    def _patch(self, idx):
        if self.buffer[idx] is None:
            if self.datapath:
                a = np.load(self.datapath[idx])
                if a.ndim != 3 or 4 not in (a.shape[0], a.shape[-1]):
                    raise ValueError(f"{self.datapath[idx]}: expected a packed raw patch (4, h, w) or (h, w, 4), got {a.shape} -- "
                                     "sRGB crops need the reference's unprocess + mosaic first (not part of this build)")
                a = a.astype(np.float32)
                self.buffer[idx] = a if a.shape[-1] == 4 else a.transpose(1, 2, 0)
            else:
                H, W = 2 * self.h, 2 * self.w
                big = S.synth_clean(4 * H, 4 * W).astype(np.float32)
                oy, ox = (idx * 37) % (3 * H) // 2 * 2, (idx * 101) % (3 * W) // 2 * 2
                self.buffer[idx] = _pack(big[oy:oy + H, ox:ox + W] * np.float32(0.4 + 0.6 * ((idx * 7) % 10) / 9.0))
        return self.buffer[idx]

    def __getitem__(self, idx, rng=None):
        """rng: the training loop's per-(epoch, rank) generator (the reference seeds numpy per loader worker)."""
        train = self.mode == 'train'
        rng = rng if rng is not None else np.random.default_rng(idx)
        data = {'name': self.names[idx] if self.names else f'synthetic_{idx:04d}'}
        hr = self._patch(idx)
        if train:
            data['aug_id1'] = int(rng.integers(8))            # drawn and unused by the reference too (:283-284 drops the result)
        data['pattern'] = 0 if 'no_bayeraug' in self.args.get('command', '') else (int(rng.integers(4)) if train else idx % 4)
        hr = np.ascontiguousarray(bayer_aug(hr, k=data['pattern']).transpose(2, 0, 1))
        if train:
            lower, upper = np.log(self.args['sigma_min']), np.log(self.args['sigma_max'])
            data['sigma'] = float(np.exp(rng.random() * (upper - lower) + lower) / 255.)
        else:
            data['sigma'] = float(self.sigma)
            rng = np.random.default_rng(idx)                     # :315 setup_seed(idx): the same noise every pass
        lr = hr + rng.standard_normal(hr.shape).astype(np.float32) * np.float32(data['sigma'])
        if self.args.get('clip', False):
            lr, hr = lr.clip(0, 1), hr.clip(0, 1)
        data['lr'], data['hr'] = lr.astype(np.float32), hr.astype(np.float32)
        return data


DIV2K_Img2Raw_Dataset = RGB_Img2Raw_Dataset                      # same item layout (yond_datasets.py:437-548), same noise model


def PSNR_Loss(low, high):
    """losses/__init__.py:3-14 for a batch already in [0, 1]: the mean over the images of -10 log10(mse)."""
    mse = ((high - low) ** 2).reshape(low.shape[0], -1).mean(dim=1)
    return float((-10.0 * torch.log(mse) / math.log(10.0)).mean())


def quality_assess(X, Y, data_range=255):
    """utils/visualization.py:26-31 for [1][C][h][w] tensors already scaled to data_range: skimage's peak_signal_noise_ratio and
    structural_similarity(channel_axis=-1) with their defaults (7 x 7 uniform window, sample covariance, K1 0.01, K2 0.03,
    the mean over the window-valid interior, averaged over the channels), restated from the published definition in float64
    -- skimage is not in this image, so the last digits are unpinned."""
    X, Y = X.double(), Y.double()
    mse = float(((X - Y) ** 2).mean())
    psnr = 10 * math.log10(data_range ** 2 / mse) if mse > 0 else float('inf')
    win, cov = 7, 49 / 48.0
    avg = lambda a: torch.nn.functional.avg_pool2d(a, win, stride=1)      # == uniform_filter on the cropped interior
    ux, uy = avg(X), avg(Y)
    vx, vy, vxy = cov * (avg(X * X) - ux * ux), cov * (avg(Y * Y) - uy * uy), cov * (avg(X * Y) - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    return {'PSNR': psnr, 'SSIM': float(s.mean())}


def load_weights(model, pretrained_dict, multi_gpu=False, by_name=False):
    """utils/utils.py:160-204: keys the model does not have (or has with another shape) are dropped with a warning."""
    model_dict = model.state_dict()
    pretrained_dict = dict(pretrained_dict)
    if by_name:
        for k in list(pretrained_dict):
            if k not in model_dict:
                log(f'Warning:  "{k}" is not exist and has been deleted!!')
                del pretrained_dict[k]
            elif model_dict[k].shape != pretrained_dict[k].shape:
                log(f'Warning:  "{k}":{pretrained_dict[k].shape}->{model_dict[k].shape}')
                del pretrained_dict[k]
    model_dict.update(pretrained_dict)
    model.load_state_dict(model_dict)
    return model


def shard_batches(n, batch_size, epoch, rank=0, world=1):
    """DistributedSampler(shuffle=True) + DataLoader(batch_size, drop_last=True) (trainer_AWGN.py:41-49): one permutation per
    epoch shared by the ranks (padded with its head to a multiple of world), rank r takes every world-th index, whole batches
    only.  Returns the list of index arrays of this rank."""
    perm = np.random.default_rng(epoch).permutation(n)
    if world > 1:
        perm = np.concatenate([perm, perm[:(-n) % world]])[rank::world]
    return [perm[b * batch_size:(b + 1) * batch_size] for b in range(len(perm) // batch_size)]


class AWGN_Parser:
    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)

    def parse(self, args=None):
        p = self.parser
        p.add_argument('--runfile', '-f', default="runfiles/Gaussian/GRU_5to50_norm_mix.yml", type=Path, help="path to config")
        p.add_argument('--mode', '-m', default='train', type=str, help="train or test")
        p.add_argument('--debug', action='store_true', default=False)
        p.add_argument('--nofig', action='store_true', default=False)
        p.add_argument('--nohost', action='store_true', default=False)
        p.add_argument('--gpu', default="0")
        p.add_argument('--local_rank', type=int, default=0)
        # not reference flags:
        p.add_argument('--synthetic', type=int, default=0, help="train / evaluate on N synthetic patches when root_dir holds none")
        p.add_argument('--epochs', type=int, default=None, help="stop after this many epochs (default: hyper.stop_epoch)")
        return p.parse_args(args)


class AWGN_Trainer:
    def __init__(self, args=None):
        self.parser = AWGN_Parser().parse(args)
        self.initialization()
        self.net = getattr(_archs, self.arch['name'])(self.arch)
        if self.hyper['last_epoch']:                              # resume: best -> last (trainer_AWGN.py:21-30)
            model_path = f'{self.fast_ckpt}/{self.model_name}_best_model.pth'
            if not os.path.exists(model_path):
                model_path = f'{self.fast_ckpt}/{self.model_name}_last_model.pth'
            if os.path.exists(model_path):
                self.net = load_weights(self.net, torch.load(model_path, map_location='cpu'), by_name=True)
            else:
                log('No checkpoint file!!!')
        else:
            log(f'Initializing {self.arch["name"]}...')
            _archs.initialize_weights(self.net)
        self.net = self.net.to(self.device)
        charb = 'gamma' in self.dst['command']                   # :64-66
        self.trainer = T.Trainer(self.net, self.hyper, charbonnier=charb)     # Adam(lr) + LambdaScheduler; DDP when a group exists
        self.ts, self.scheduler = self.trainer.ts, self.trainer.scheduler
        self.print_model_log()
        if self.mode == 'train':
            self.dst_train = globals()[self.args['dst_train']['dataset']](self.args['dst_train'], self.parser.synthetic)
        self.change_eval_dst('eval')

    # -- trainer_base.py:48-82 ------------------------------------------------------------------------------------------------
    def initialization(self):
        with open(self.parser.runfile, 'r', encoding='utf-8') as f:
            self.args = yaml.load(f.read(), Loader=yaml.FullLoader)
        self.mode = self.args['mode'] if self.parser.mode is None else self.parser.mode
        if 'clip' not in self.args['dst']:
            self.args['dst']['clip'] = False
        self.args['dst']['mode'] = self.mode
        self.rank, self.local_rank, self.world = D.init()
        self.multi_gpu = self.world > 1
        if not torch.cuda.is_available():
            raise SystemExit("trainer_AWGN needs an MI355X: the HIP path has no CPU fallback")
        self.device = torch.device('cuda', self.local_rank)
        torch.cuda.set_device(self.device)
        self.dst, self.hyper, self.arch = self.args['dst'], self.args['hyper'], self.args['arch']
        self.model_name, self.fast_ckpt, self.model_dir = self.args['model_name'], self.args['fast_ckpt'], self.args['checkpoint']
        self.sample_dir = os.path.join(self.args['result_dir'], f"samples-{self.model_name}")
        for d in (self.model_dir, self.sample_dir, './logs', f'./{self.fast_ckpt}', './metrics'):
            os.makedirs(d, exist_ok=True)

    def print_model_log(self):
        self.best_psnr = self.hyper['best_psnr'] if 'best_psnr' in self.hyper else 0
        last_eval_epoch = self.hyper['last_epoch'] // self.hyper['plot_freq']
        self.train_psnr = AverageMeter('PSNR', last_epoch=self.hyper['last_epoch'])
        self.eval_psnr = AverageMeter('PSNR', last_epoch=last_eval_epoch)
        self.eval_ssim = AverageMeter('SSIM', last_epoch=last_eval_epoch)
        self.logfile = f'./logs/log_{self.model_name}.log'
        if self.rank == 0:
            for k, v in (('Model Name', self.model_name), ('Architecture', self.arch['name']),
                         ('TrainDataset', self.args['dst_train']['dataset']), ('EvalDataset', self.args['dst_eval']['dataset']),
                         ('num_channels', self.arch['nf']), ('BatchSize', self.hyper['batch_size']), ('PatchSize', self.dst['patch_size']),
                         ('LearningRate', self.hyper['learning_rate']), ('Epoch', self.hyper['stop_epoch']), ('Command', self.dst['command'])):
                log(f'{k}:\t{v}', log=self.logfile, notime=True)
            log(f"Let's use {self.world} GPUs (one process each, gradients averaged over {'RCCL' if self.multi_gpu else 'nothing'})!",
                log=self.logfile, notime=True)

    def change_eval_dst(self, mode='eval'):
        self.dst = self.args[f'dst_{mode}']
        self.dstname = self.dst['dstname']
        self.dst_eval = globals()[self.dst['dataset']](self.dst, self.parser.synthetic)

    # -- trainer_AWGN.py:347-368 -----------------------------------------------------------------------------------------------
    def preprocess(self, data, mode='train', preprocess=True):
        imgs_hr = torch.as_tensor(np.asarray(data['hr']), dtype=torch.float32).to(self.device)
        imgs_lr = torch.as_tensor(np.asarray(data['lr']), dtype=torch.float32).to(self.device)
        imgs_hr = imgs_hr.reshape(-1, *imgs_hr.shape[-3:])
        imgs_lr = imgs_lr.reshape(-1, *imgs_lr.shape[-3:])
        sigma = torch.as_tensor(np.asarray(data['sigma']), dtype=torch.float32).to(self.device).view(-1, 1, 1, 1)
        if self.dst['clip']:
            imgs_lr, imgs_hr = imgs_lr.clamp(0, 1), imgs_hr.clamp(0, 1)
        return imgs_lr, imgs_hr, sigma

    def _epoch_batches(self, epoch):
        rng = np.random.default_rng([epoch, self.rank, 1997])
        for idx in shard_batches(len(self.dst_train), self.hyper['batch_size'], epoch, self.rank, self.world):
            items = [self.dst_train.__getitem__(int(i), rng) for i in idx]
            yield {k: (np.stack([it[k] for it in items]) if k in ('lr', 'hr') else [it[k] for it in items]) for k in items[0]}

    def state_dict(self):
        return {k: v.detach().cpu() for k, v in self.net.state_dict().items()}

    # -- trainer_AWGN.py:78-193 ------------------------------------------------------------------------------------------------
    def train(self):
        pf = self.hyper['plot_freq']
        lr = self.scheduler.get_last_lr()[0]
        first = self.hyper['last_epoch'] + 1
        last = self.hyper['stop_epoch'] if self.parser.epochs is None else min(self.hyper['stop_epoch'], first + self.parser.epochs - 1)
        guided = 'guided' in self.args['arch']
        for epoch in range(first, last + 1):
            D.barrier()
            self.train_psnr.reset()
            t0, nb, losses = time.perf_counter(), 0, []
            for data in self._epoch_batches(epoch):
                imgs_lr, imgs_hr, sigma = self.preprocess(data, mode='train', preprocess=True)
                losses.append(self.ts.step(imgs_lr, imgs_hr, sigma if guided else None)[0])
                with torch.no_grad():                           # :120-124 on the step's own prediction
                    self.train_psnr.update(PSNR_Loss(self.ts.last_pred.clamp(0, 1), imgs_hr.clamp(0, 1)))
                nb += 1
            self.scheduler.step()
            self.trainer.history.append((epoch, lr, losses))
            D.barrier()
            if self.rank == 0:
                log(f'Epoch {epoch}: lr={lr:.2e}, {nb} batches of {self.hyper["batch_size"]} per rank, loss={np.mean(losses):.5f}, '
                    f'PSNR={self.train_psnr.avg:.2f}, {time.perf_counter() - t0:.2f} s', log=self.logfile)
                if epoch % self.hyper['save_freq'] == 0:
                    sd = self.state_dict()
                    _atomic_save(sd, os.path.join(self.model_dir, '%s_e%04d.pth' % (self.model_name, epoch // pf * pf)))
                    _atomic_save(sd, f'{self.fast_ckpt}/{self.model_name}_last_model.pth')
                if epoch % pf == 0:                             # fast eval
                    log(f"learning_rate: {lr:.3e}")
                    self.dst_eval.sigma = self.args['dst_eval']['sigma_list'][1] / 255.
                    self.eval(epoch=epoch)
                    _atomic_save(self.state_dict(), f'{self.fast_ckpt}/{self.model_name}_last_model.pth')
            lr = self.scheduler.get_last_lr()[0]
        return self.trainer.history

    # -- trainer_AWGN.py:195-310 -----------------------------------------------------------------------------------------------
    def eval(self, epoch=-1):
        self.train_psnr.reset(), self.eval_psnr.reset(), self.eval_ssim.reset()
        metrics = {}
        metrics_path = f'./metrics/{self.model_name}_metrics.pkl'
        if os.path.exists(metrics_path):
            with open(metrics_path, 'rb') as f:
                metrics = pkl.load(f)
        guided = 'guided' in self.args['arch']
        for k in range(len(self.dst_eval)):
            data = self.dst_eval[k]
            imgs_lr, imgs_hr, sigma = self.preprocess({'lr': data['lr'][None], 'hr': data['hr'][None], 'sigma': data['sigma']},
                                                      mode='eval', preprocess=False)
            name = data['name'] + f'_sig{int(sigma.item() * 255)}'
            pad = imgs_lr.shape[-1] % 16 != 0                  # :220-225 (the reference calls a guided net without sigma there)
            x = torch.nn.functional.pad(imgs_lr, (4, 4, 4, 4), mode='reflect') if pad else imgs_lr
            imgs_dn = self.net(x, sigma) if guided else self.net(x)
            imgs_dn = imgs_dn[..., 4:-4, 4:-4] if pad else imgs_dn
            res = quality_assess(imgs_dn.clamp(0, 1) * 255.0, (imgs_hr * 255.0).clamp(0, 255), data_range=255)
            self.eval_psnr.update(res['PSNR'])
            self.eval_ssim.update(res['SSIM'])
            metrics[name] = [res['PSNR'], res['SSIM']]
        if self.eval_psnr.avg >= self.best_psnr and epoch > 0:
            self.best_psnr = self.eval_psnr.avg
            log(f"Best PSNR is {self.best_psnr} now!!")
            _atomic_save(self.state_dict(), f'{self.fast_ckpt}/{self.model_name}_best_model.pth')
        log(f"Epoch {epoch}: PSNR={self.eval_psnr.avg:.2f}, SSIM={self.eval_ssim.avg:.4f}", log=self.logfile)
        if epoch < 0:
            with open(metrics_path, 'wb') as f:
                pkl.dump(metrics, f)
        return metrics


def _atomic_save(obj, path):
    """torch.save to a temporary file beside `path`, then os.replace: a reader (another rank, a resumed run) never sees a
    half-written checkpoint."""
    tmp = f"{path}.tmp{os.getpid()}"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def main(argv=None):
    """trainer_AWGN.py:382-404."""
    trainer = AWGN_Trainer(argv)
    try:
        out = {}
        if trainer.mode == 'train':
            out['history'] = trainer.train()
            D.barrier()                     # rank 0 may still be inside the last epoch's save / eval block: no rank loads before it is done
            trainer.mode = 'evaltest'
        best = f'{trainer.fast_ckpt}/{trainer.model_name}_best_model.pth'
        if not os.path.exists(best):
            best = f'{trainer.fast_ckpt}/{trainer.model_name}_last_model.pth'
        if os.path.exists(best):
            trainer.net = load_weights(trainer.net, torch.load(best, map_location=trainer.device), by_name=True)
        if 'eval' in trainer.mode and trainer.rank == 0:
            trainer.change_eval_dst('eval')
            for sigma in trainer.args['dst_test']['sigma_list']:
                log(f'AWGN Datasets: sigma={sigma}', log=trainer.logfile)
                trainer.dst_eval.sigma = sigma / 255.
                out[f'metrics_sig{sigma}'] = dict(trainer.eval(-1))
                out[f'psnr_sig{sigma}'], out[f'ssim_sig{sigma}'] = trainer.eval_psnr.avg, trainer.eval_ssim.avg
            log(f'Metrics have been saved in ./metrics/{trainer.model_name}_metrics.pkl')
        D.barrier()
        if D._active() and trainer.rank == 0:
            import json
            print("DIST_STATS " + json.dumps(D.STATS), flush=True)
        return out
    finally:
        D.finalize()


if __name__ == '__main__':
    main()
