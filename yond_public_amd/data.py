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


# ------------------------------------------------------------------------------------------------
# Full-frame datasets: the reference reads DNG / ARW / CR2 / NEF through rawpy, which this image
# lacks, so every class below works on `.npy`-CONVERTED trees: the same directory layout and item dictionaries, each raw
# file replaced by (or accompanied by) `<same name>.npy` holding raw_image_visible as the camera wrote it.  Black / white
# level come from the camera directory's `meta.json` ({"bl": .., "wp": ..}: what rawpy's black_level_per_channel /
# white_level report) or, without one, from the runfile's dst section.
# ------------------------------------------------------------------------------------------------
def _npy_path(path):
    """The converted copy of a raw file: `<stem>.npy` next to it (or the path itself when it already is one)."""
    if path.endswith('.npy'):
        return path
    stem = os.path.splitext(path)[0]
    if os.path.exists(stem + '.npy'):
        return stem + '.npy'
    raise RuntimeError(f"{path}: reading {os.path.splitext(path)[1] or 'this'} files needs rawpy, which this build does not have; "
                       f"convert it once (np.save('{stem}.npy', rawpy.imread(path).raw_image_visible)) -- the datasets pick up "
                       f"`.npy` copies next to the raw files")


def _levels(dirpath, args):
    import json
    f = os.path.join(dirpath, 'meta.json')
    if os.path.exists(f):
        m = json.load(open(f))
        return float(m.get('bl', args['bl'])), float(m.get('wp', args['wp']))
    return float(args['bl']), float(args['wp'])


class ELD_Full_Dataset:
    """data_process/yond_datasets.py:977-1067: <root_dir>/<camera>/scene-<1..10>/IMG_<id>.<suffix>, 3 ISOs x 4 ratios per
    scene, long-exposure references IMG_0001 / 0006 / 0011 / 0016; `change_eval_ratio(cam, ratio, iso_list)` selects the
    evaluated subset.  Items: 'lr' = (raw - bl) * ratio / (wp - bl), 'hr' = (raw - bl) / (wp - bl) (float32 Bayer frames),
    'name', 'ratio', 'ISO', 'wb', 'ccm' (identity without the raw file's metadata)."""
    SUFFIX = {'CanonEOS70D': 'CR2', 'CanonEOS700D': 'CR2', 'NikonD850': 'nef', 'SonyA7S2': 'ARW'}

    def __init__(self, args=None):
        self.args = {'root_dir': 'ELD/', 'ratio': 1, 'dstname': 'ELD', 'params': None, 'mode': 'eval', 'command': '',
                     'wp': 16383, 'bl': 512, 'clip': False}
        self.args.update(args or {})
        self.infos_all = {cam: [] for cam in self.SUFFIX}
        iso_list, ratio_list = [800, 1600, 3200], [1, 10, 100, 200]
        hr_ids = np.array([1, 6, 11, 16])
        for cam, suffix in self.SUFFIX.items():
            sub_dir = f'{self.args["root_dir"]}/{cam}'
            for scene in range(1, 11):
                for iso_id, iso in enumerate(iso_list):
                    for ratio_id, ratio in enumerate(ratio_list):
                        lr_id = iso_id * 5 + ratio_id + 2                                   # :1012
                        hr_id = hr_ids[np.argmin(np.abs(lr_id - hr_ids))]
                        self.infos_all[cam].append({
                            'cam': cam, 'name': f'{cam}_{scene:02d}_IMG_{lr_id:04d}',
                            'hr': f'{sub_dir}/scene-{scene}/IMG_{hr_id:04d}.{suffix}',
                            'lr': f'{sub_dir}/scene-{scene}/IMG_{lr_id:04d}.{suffix}', 'iso': iso, 'ratio': ratio})
        self.change_eval_ratio('SonyA7S2', ratio=1)

    def __len__(self):
        return self.length

    def change_eval_ratio(self, cam='SonyA7S2', ratio=1, iso_list=None):
        iso_list = iso_list or [800, 1600, 3200]
        # (scenes that are not on disk are skipped: a converted subset of the dataset is a valid tree)
        self.infos = [i for i in self.infos_all[cam] if i['iso'] in iso_list and i['ratio'] == ratio
                      and (os.path.exists(i['lr']) or os.path.exists(os.path.splitext(i['lr'])[0] + '.npy'))]
        self.length, self.ratio, self.cam = len(self.infos), ratio, cam
        self.bl, self.wp = _levels(f'{self.args["root_dir"]}/{cam}', self.args)
        if self.infos:
            self.H, self.W = np.load(_npy_path(self.infos[0]['lr']), mmap_mode='r').shape

    def __getitem__(self, idx):
        info = self.infos[idx]
        hr_raw = np.load(_npy_path(info['hr'])).reshape(self.H, self.W)
        lr_raw = np.load(_npy_path(info['lr'])).reshape(self.H, self.W)
        data = {'hr': (hr_raw.astype(np.float32) - self.bl) / (self.wp - self.bl),
                'lr': (lr_raw.astype(np.float32) - self.bl) * info['ratio'] / (self.wp - self.bl),
                'name': info['name'], 'ratio': info['ratio'], 'ISO': info['iso'], 'cfa': 'rggb',
                'wb': np.ones(4, np.float32), 'ccm': np.eye(3, dtype=np.float32), 'meta': None}
        data['hr'], data['lr'] = data['hr'].astype(np.float32), data['lr'].astype(np.float32)
        if self.args['clip']:
            data['hr'], data['lr'] = data['hr'].clip(0, 1), data['lr'].clip(0, 1)
        return data


class LRID_Dataset:
    """data_process/yond_datasets.py:870-975.  The reference lists its files in pickled info tables (`infos/<dstname>_<GT_type>.info`,
    `infos/<dstname>_short.info`, not shipped) whose entries point at DNG files; here the same tables are read when present
    (paths may name the DNG: the `.npy` copy next to it is loaded), else the converted tree
        <root_dir>/<dstname>/<scene id>/gt.npy, x<ratio:02d>.npy [, meta.json {"wb": [4], "ccm": [3][3], "ExposureTime": s}]
    is scanned; `get_eval_id` keeps the reference's evaluation scenes (:938-950).  Items as the reference's: 'lr' scaled by
    the ratio, 'hr', 'name' = '<scene>_x<ratio:02d>', 'ratio', 'wb', 'ccm', 'ISO' = 6400, 'ExposureTime' in ms."""

    def __init__(self, args=None):
        self.args = {'root_dir': 'LRID/', 'suffix': 'dng', 'dgain': 1, 'dstname': ['indoor_x5'], 'camera_type': 'IMX686', 'params': None,
                     'mode': 'eval', 'GT_type': 'GT_align_ours', 'command': '', 'H': 3472, 'W': 4624, 'wp': 1023, 'bl': 64, 'clip': False}
        self.args.update(args or {})
        if isinstance(self.args['dstname'], str):
            self.args['dstname'] = [self.args['dstname']]
        self.iso = 6400
        self.change_eval_ratio(ratio=1)

    def __len__(self):
        return self.length

    @staticmethod
    def get_eval_id(dstname='indoor_x5'):
        return {'indoor_x5': [4, 14, 25, 41, 44, 51, 52, 53, 58], 'indoor_x3': [], 'outdoor_x5': [1, 2, 5],
                'outdoor_x3': [9, 21, 22, 32, 44, 51]}.get(dstname, [])

    def change_eval_ratio(self, ratio):
        import json
        import pickle as pkl
        self.ratio, self.infos = ratio, []
        for dstname in self.args['dstname']:
            gt_info, short_info = f"infos/{dstname}_{self.args['GT_type']}.info", f'infos/{dstname}_short.info'
            ids = self.get_eval_id(dstname)
            if os.path.exists(gt_info) and os.path.exists(short_info):           # the reference's tables (:917-931)
                with open(gt_info, 'rb') as f:
                    gts = pkl.load(f)
                with open(short_info, 'rb') as f:
                    shorts = pkl.load(f)[ratio]
                for i in ids:
                    self.infos.append({'name': gts[i]['name'], 'hr': gts[i]['data'], 'lr': shorts[i]['data'][0], 'wb': gts[i]['wb'],
                                       'ccm': gts[i]['ccm'], 'ExposureTime': shorts[i]['metadata'][0]['ExposureTime']})
                continue
            base = f"{self.args['root_dir']}/{dstname}"
            for i in ids:                                                        # the converted tree
                d = f'{base}/{i:03d}'
                lr = f'{d}/x{ratio:02d}.npy'
                if not os.path.exists(lr):
                    continue
                meta = json.load(open(f'{d}/meta.json')) if os.path.exists(f'{d}/meta.json') else {}
                self.infos.append({'name': f'{dstname}_{i:03d}', 'hr': f'{d}/gt.npy', 'lr': lr,
                                   'wb': np.asarray(meta.get('wb', [1, 1, 1, 1]), np.float32),
                                   'ccm': np.asarray(meta.get('ccm', np.eye(3)), np.float32),
                                   'ExposureTime': float(meta.get('ExposureTime', 0.0))})
        self.length = len(self.infos)
        self.bl, self.wp = _levels(self.args['root_dir'], self.args)
        self.H, self.W = self.args['H'], self.args['W']
        if self.infos:
            self.H, self.W = np.load(_npy_path(self.infos[0]['lr']), mmap_mode='r').shape

    def __getitem__(self, idx):
        info = self.infos[idx]
        hr_raw = np.load(_npy_path(info['hr'])).reshape(self.H, self.W)
        lr_raw = np.load(_npy_path(info['lr'])).reshape(self.H, self.W)
        data = {'hr': ((hr_raw.astype(np.float32) - self.bl) / (self.wp - self.bl)).astype(np.float32),
                'lr': ((lr_raw.astype(np.float32) - self.bl) * self.ratio / (self.wp - self.bl)).astype(np.float32),
                'name': f"{info['name']}_x{self.ratio:02d}", 'ratio': self.ratio, 'ccm': info['ccm'], 'wb': info['wb'], 'cfa': 'rggb',
                'ISO': self.iso, 'ExposureTime': info['ExposureTime'] * 1000, 'meta': None}
        if self.args['clip']:
            data['hr'], data['lr'] = data['hr'].clip(0, 1), data['lr'].clip(0, 1)
        return data


class Any_Dataset:
    """README.md:38-47 'YOND_any': any directory of Bayer frames.  <root_dir>/*.npy (raw DN as the camera wrote them; a
    matching <root_dir>/gt/<same name>.npy is used as the reference frame when present); black / white level, ratio and clip
    from the runfile's dst section.  Items: 'lr' = (raw - bl) * ratio / (wp - bl), 'name', 'ratio'."""

    def __init__(self, args=None):
        self.args = {'root_dir': 'frames/', 'wp': 1023, 'bl': 64, 'clip': False, 'mode': 'eval', 'dstname': 'ANY'}
        self.args.update(args or {})
        self.bl, self.wp = _levels(self.args['root_dir'], self.args)
        self.files = sorted(glob.glob(os.path.join(self.args['root_dir'], '*.npy')))
        self.change_eval_ratio(1)

    def __len__(self):
        return len(self.files)

    def change_eval_ratio(self, ratio):
        self.ratio = ratio

    def __getitem__(self, idx):
        f = self.files[idx]
        raw = np.load(f).astype(np.float32)
        data = {'lr': ((raw - self.bl) * self.ratio / (self.wp - self.bl)).astype(np.float32),
                'name': f"{os.path.splitext(os.path.basename(f))[0]}_x{self.ratio:02d}", 'ratio': self.ratio, 'cfa': 'rggb', 'meta': None}
        g = os.path.join(self.args['root_dir'], 'gt', os.path.basename(f))
        if os.path.exists(g):
            data['hr'] = ((np.load(g).astype(np.float32) - self.bl) / (self.wp - self.bl)).astype(np.float32)
        if self.args['clip']:
            data['lr'] = data['lr'].clip(0, 1)
            if 'hr' in data:
                data['hr'] = data['hr'].clip(0, 1)
        return data


# ------------------------------------------------------------------------------------------------
# Feeding the GPU (YOND_SIDD.py:507-514 reads every item in front of its IterDenoise call: tens of milliseconds of file reads and a
# 64 MB upload against a few milliseconds of GPU time per image).
SIDD_FRAME_HW = {'S6': (3000, 5328), 'GP': (3044, 4048), 'N6': (3120, 4208), 'G4': (2988, 5312), 'IP': (3024, 4032)}   # the five phones of the SIDD scenes


def item_sizes(ds):
    """Pixel counts of the items' full frames WITHOUT loading them (what the size-aware sharding needs before launch):
    a dataset's own `item_size(k)`, the header of a `.npy` full frame, the phone code in a SIDD scene name (0009_001_S6_...);
    None where nothing is known (the caller then shards round-robin)."""
    out = []
    for k in range(len(ds)):
        n = None
        if hasattr(ds, 'item_size'):
            n = ds.item_size(k)
        elif hasattr(ds, 'infos'):
            info = ds.infos[k]
            path, name = info.get('lr_path'), str(info.get('name', ''))
            if path and path.endswith('.npy') and os.path.exists(path):
                shp = np.load(path, mmap_mode='r').shape
                n = int(shp[-1]) * int(shp[-2])
            else:
                parts = name.split('_')
                if len(parts) > 2 and parts[2] in SIDD_FRAME_HW:
                    n = SIDD_FRAME_HW[parts[2]][0] * SIDD_FRAME_HW[parts[2]][1]
        if n is None:
            return None
        out.append(int(n))
    return out


class _PinnedPool:
    """Pinned staging buffers shared by a Prefetcher's workers: (key, shape) -> buffers with the event of their last upload.  A buffer is handed
    out again once that event has completed; buffers of shapes that have not been used lately are dropped when the pool would grow past `budget`
    bytes (five SIDD frame sizes x per-worker double buffers had pinned ~2.5 GB of host memory)."""

    def __init__(self, budget=1 << 30):
        import threading
        self.lock, self.bufs, self.bytes, self.budget, self.clock = threading.Lock(), {}, 0, int(budget), 0

    def take(self, key, shape):
        import torch
        pk = (key, tuple(shape))
        with self.lock:
            self.clock += 1
            lst = self.bufs.setdefault(pk, [])
            for b in lst:
                if not b['busy'] and (b['ev'] is None or b['ev'].query()):
                    b['busy'], b['used'] = True, self.clock
                    return b
            need = 4
            for d in shape:
                need *= int(d)
            if self.bytes + need > self.budget:                             # drop idle buffers of other shapes, least recently used first
                idle = sorted(((b['used'], k2, b) for k2, l2 in self.bufs.items() if k2 != pk for b in l2
                               if not b['busy'] and (b['ev'] is None or b['ev'].query())), key=lambda t: t[0])
                for _, k2, b in idle:
                    if self.bytes + need <= self.budget:
                        break
                    self.bufs[k2].remove(b)
                    self.bytes -= b['t'].numel() * 4
            b = {'t': torch.empty(tuple(shape), dtype=torch.float32).pin_memory(), 'ev': None, 'busy': True, 'used': self.clock}
            lst.append(b)
            self.bytes += need
            return b

    def give(self, b, ev):
        with self.lock:
            b['ev'], b['busy'] = ev, False


class Prefetcher:
    """`for k, data in Prefetcher(ds, indices, device)`: the items `ds[k]` in the order of `indices`, read by `workers` background
    threads up to `depth` items ahead; every numpy array under the keys of `upload` arrives as a float32 DEVICE tensor (copied into
    a reused pinned buffer of a pool the workers share, sent on the worker's own copy stream; the consumer's current stream is made to wait
    for the copy's event, so no host synchronisation happens on the consumer's side).  Everything else in the item is passed through.  An
    exception in a worker is re-raised at the item it belongs to.
    The look-ahead is gated by POSITION: the worker of position p starts once p < consumed + depth.  (A counting semaphore lets workers that
    have already delivered take every permit for positions beyond the one the consumer is blocked on, while that position's worker starves.)"""

    def __init__(self, ds, indices, device, upload=('lr', 'hr', 'lr_full'), depth=4, workers=4, pinned_budget=1 << 30):
        import queue
        import threading
        import torch
        self.ds, self.indices, self.device, self.upload = ds, list(indices), torch.device(device), tuple(upload)
        self.workers = max(1, min(int(workers), len(self.indices) or 1))
        self.depth = max(self.workers, int(depth))
        self._slots = [queue.Queue(maxsize=1) for _ in self.indices]        # one-shot mailboxes, filled out of order, read in order
        self._cv = threading.Condition()
        self._consumed = 0                                                  # positions the consumer has taken
        self._stop = False
        self._pool = _PinnedPool(pinned_budget)
        self._threads = [threading.Thread(target=self._work, args=(w,), daemon=True) for w in range(self.workers)]
        for t in self._threads:
            t.start()

    def _work(self, w):
        import torch
        stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        for pos in range(w, len(self.indices), self.workers):
            with self._cv:
                self._cv.wait_for(lambda: self._stop or pos < self._consumed + self.depth)
                if self._stop:
                    return
            k = self.indices[pos]
            try:
                data = dict(self.ds[k])
                ev = None
                for key in self.upload:
                    a = data.get(key)
                    if not isinstance(a, np.ndarray):
                        continue
                    if stream is None:
                        data[key] = torch.from_numpy(np.ascontiguousarray(a, np.float32))
                        continue
                    buf = self._pool.take(key, a.shape)
                    buf['t'].numpy()[...] = a                               # (converts to float32 on the way)
                    with torch.cuda.stream(stream):
                        data[key] = buf['t'].to(self.device, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                    self._pool.give(buf, ev)
                self._slots[pos].put((k, data, ev, None))
            except BaseException as e:                                      # noqa: BLE001 -- handed to the consumer
                self._slots[pos].put((k, None, None, e))

    def __iter__(self):
        import torch
        try:
            for pos in range(len(self.indices)):
                k, data, ev, err = self._slots[pos].get()
                with self._cv:
                    self._consumed = pos + 1
                    self._cv.notify_all()
                if err is not None:
                    raise err
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for key in self.upload:
                        if isinstance(data.get(key), torch.Tensor) and data[key].is_cuda:
                            data[key].record_stream(cur)
                yield k, data
        finally:
            with self._cv:
                self._stop = True
                self._cv.notify_all()
