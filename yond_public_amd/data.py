"""Dataset ingestion for the evaluation entry point: the reference's SIDD_Dataset
(data_process/yond_datasets.py:767-868) and dataload (utils/utils.py:319-335) for the formats this image can read --
.npy and MATLAB v5 .mat (scipy.io); v7.3 .mat needs h5py and DNG / ARW rawpy, neither of which is installed: those raise
a clear error instead of guessing.  Items are host arrays (the pipeline uploads them)."""
import glob
import os

import numpy as np


def dataload(path):
    """utils/utils.py:319-335 (the suffixes readable here)."""
    suffix = path[-4:].lower()
    if suffix == '.npy':
        return np.load(path)
    if suffix == '.raw':
        return np.fromfile(path, np.uint16).reshape(1440, 2560)
    if suffix == '.mat':
        import scipy.io
        if 'metadata' in path.lower():
            return scipy.io.loadmat(path)
        try:
            import h5py                                   # the full-resolution SIDD frames are MATLAB v7.3 files (:331-332)
        except ImportError:
            try:
                # a v5 re-save of the same variable.  h5py hands MATLAB's column-major array over with reversed axes (the
                # reference works on that orientation, utils/utils.py:331-332); loadmat returns MATLAB's own: transpose
                return np.ascontiguousarray(np.asarray(scipy.io.loadmat(path)['x']).T)
            except NotImplementedError as e:
                raise RuntimeError(f"{path} is a MATLAB v7.3 file and h5py is not installed; convert it to .npy "
                                   f"(np.save of f['x']) and point lr_path_full at that") from e
        with h5py.File(path, 'r') as f:
            return np.array(f['x'])
    raise RuntimeError(f"dataload: unsupported file type {path!r} (rawpy / cv2 formats are outside this build)")


class SIDD_Dataset:
    """data_process/yond_datasets.py:767-868, eval / test modes: the 40 x 32 x 256 x 256 validation blocks
    (SIDD_Validation_Raw/Validation{Noisy,Gt}BlocksRaw.mat, MATLAB v5) plus, when SIDD_Benchmark_Data is present, the
    per-scene metadata (CFA pattern, ISO, the camera's noise model) and the path of the full-resolution noisy frame that
    round 1 estimates from (YOND_SIDD.py:339-341)."""

    def __init__(self, args=None):
        self.args = {'root_dir': '/data/fenghansen/datasets/SIDD', 'mode': 'eval', 'dstname': 'SIDD', 'wp': 1023, 'bl': 64,
                     'patch_size': 256, 'H': 256, 'W': 256, 'clip': True, 'command': '', 'lock_wb': False, 'params': None,
                     'gpu_preprocess': False}
        self.args.update(args or {})
        self.root_dir, self.mode = self.args['root_dir'], self.args['mode']
        import scipy.io as sio
        vr = f'{self.root_dir}/SIDD_Validation_Raw'
        if self.mode == 'eval':
            self.lr_data = sio.loadmat(f'{vr}/ValidationNoisyBlocksRaw.mat')['ValidationNoisyBlocksRaw']
            self.hr_data = sio.loadmat(f'{vr}/ValidationGtBlocksRaw.mat')['ValidationGtBlocksRaw']
        else:
            self.lr_data = sio.loadmat(f'{vr}/BenchmarkNoisyBlocksRaw.mat')['BenchmarkNoisyBlocksRaw']
            self.hr_data = None
        self.data_dir = f'{self.root_dir}/SIDD_Benchmark_Data'
        self.infos = []
        if os.path.isdir(self.data_dir):
            from .utils.sidd_utils import read_metadata
            names = sorted(os.listdir(self.data_dir))
            paths = sorted(glob.glob(f'{self.data_dir}//*/*_010.MAT'))
            metas = sorted(q for q in paths if 'META' in q)
            lrs = sorted(q for q in paths if 'NOISY' in q)
            for i, name in enumerate(names[:len(self.lr_data)]):
                md = read_metadata(dataload(metas[i])) if i < len(metas) else None
                full = lrs[i] if i < len(lrs) else None
                npy = full[:-4] + '.npy' if full else None              # a converted copy next to the v7.3 file is preferred
                self.infos.append({'name': name, 'lr_path': npy if npy and os.path.exists(npy) else full, 'metadata': md})
        while len(self.infos) < len(self.lr_data):
            self.infos.append({'name': f'sidd_{len(self.infos):04d}', 'lr_path': None, 'metadata': None})
        self.length = len(self.lr_data)

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        info = self.infos[idx]
        md = info['metadata']
        data = {'name': info['name'], 'meta': md, 'lr_path_full': info['lr_path'],
                'cfa': md['bayer_2by2'] if md else [[1, 2], [2, 3]], 'lr': np.asarray(self.lr_data[idx], np.float32)}
        if md:
            data.update(wb=md['wb'], ccm=md['cst2'], iso=md['iso'], reg=(md['beta1'], md['beta2']))
        if self.hr_data is not None:
            data['hr'] = np.asarray(self.hr_data[idx], np.float32)
        data['lr_full'] = None
        if info['lr_path'] is not None:
            try:
                data['lr_full'] = np.asarray(dataload(info['lr_path']), np.float32)
            except RuntimeError as e:                                       # v7.3 without h5py: estimate from the blocks (:340)
                data['lr_full_error'] = str(e)
        return data
