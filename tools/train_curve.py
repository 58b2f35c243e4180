"""End-to-end sanity of the training stack: the trainer_AWGN driver on synthetic patches with a small network, long enough to see the
evaluation PSNR climb (python tools/train_curve.py [LR] [EPOCHS] [NF]).  Prints the fast-eval lines (sigma = 25) and the final
evaluation over sigma 10 / 25 / 50."""
import os
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from yond_public_amd import trainer_AWGN as TA

lr = float(sys.argv[1]) if len(sys.argv) > 1 else 5e-3
ep = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tmp = tempfile.mkdtemp()
os.chdir(tmp)
cfg = yaml.load(open(os.path.join(ROOT, "runfiles", "Gaussian", "GRU_5to50_norm_mix.yml")).read(), Loader=yaml.FullLoader)
cfg["arch"]["nf"] = nf
for sec in ("dst", "dst_train", "dst_eval", "dst_test"):
    cfg[sec].update(H=64, W=64, patch_size=64, root_dir=tmp + "/nodata")
cfg["hyper"].update(batch_size=8, last_epoch=0, stop_epoch=ep, step_size=2, T=1, coldstart=True, save_freq=10 ** 6, plot_freq=max(ep // 10, 1),
                    learning_rate=lr)
open("rf.yml", "w").write(yaml.dump(cfg))
torch.manual_seed(11)
TA.main(['-f', 'rf.yml', '-m', 'train', '--synthetic', '64'])
