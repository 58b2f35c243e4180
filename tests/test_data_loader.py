"""CPU: dataset ingestion (N3) -- dataload and SIDD_Dataset on a miniature tree in the reference's layout
(data_process/yond_datasets.py:767-868, utils/utils.py:319-335), written here with scipy.io.savemat (MATLAB v5)."""
import os

import numpy as np
import pytest
import scipy.io as sio


def make_tree(root, n=2, with_meta=True):
    rng = np.random.default_rng(0)
    vr = root / "SIDD_Validation_Raw"
    vr.mkdir(parents=True)
    lr = rng.random((n, 32, 8, 8)).astype(np.float32)
    hr = rng.random((n, 32, 8, 8)).astype(np.float32)
    sio.savemat(vr / "ValidationNoisyBlocksRaw.mat", {"ValidationNoisyBlocksRaw": lr})
    sio.savemat(vr / "ValidationGtBlocksRaw.mat", {"ValidationGtBlocksRaw": hr})
    fulls = []
    if with_meta:
        for i in range(n):
            d = root / "SIDD_Benchmark_Data" / f"{i:04d}_00{i}_GP_00100_00060_3200_L"
            d.mkdir(parents=True)
            # the metadata struct, indexed as utils/sidd_utils.py:3-73 index it
            tags = np.empty((8, 1), dtype=object)
            for t in range(8):
                tags[t, 0] = (np.array([[0]]), np.array([[0]]), np.array([[0.0, 0.0]]))
            tags[7, 0] = (np.array([[0]]), np.array([[0]]), np.array([[1e-3 * (i + 1), 2e-6]]))
            unknown = np.zeros((8, 1), dtype=[('ID', object), ('Type', object), ('Value', object)])
            for t in range(8):
                unknown[t, 0] = (np.array([[0]]), np.array([[0]]), np.array([[0.0, 0.0]]))
            unknown[1, 0] = (np.array([[33422]]), np.array([[1]]), np.array([[1, 0, 2, 1]]))        # GRBG
            unknown[7, 0] = (np.array([[51041]]), np.array([[12]]), np.array([[1e-3 * (i + 1), 2e-6]]))
            meta = {'UnknownTags': unknown, 'Make': np.array(['Google']), 'AsShotNeutral': np.array([[0.5, 1.0, 0.6]]),
                    'ColorMatrix1': np.arange(9.0).reshape(1, 9), 'ColorMatrix2': np.arange(9.0).reshape(1, 9) + 1,
                    'ISOSpeedRatings': np.array([[100 * (i + 1)]])}
            sio.savemat(d / f"{i:04d}_METADATA_RAW_010.MAT", {"metadata": meta})
            full = rng.random((16, 24)).astype(np.float32)
            np.save(d / f"{i:04d}_NOISY_RAW_010.npy", full)
            # stands for the v7.3 file (the .npy copy is preferred): a v5 re-save FROM MATLAB holds the variable in MATLAB's
            # orientation, the transpose of what the reference's h5py read of the v7.3 file yields (utils/utils.py:331-332)
            sio.savemat(d / f"{i:04d}_NOISY_RAW_010.MAT", {"x": np.ascontiguousarray(full.T)})
            fulls.append(full)
    return lr, hr, fulls


def test_sidd_dataset_reads_the_reference_layout(tmp_path):
    from yond_public_amd.data import SIDD_Dataset, dataload
    lr, hr, fulls = make_tree(tmp_path)
    ds = SIDD_Dataset({'root_dir': str(tmp_path), 'mode': 'eval'})
    assert len(ds) == 2
    for i in range(2):
        d = ds[i]
        assert np.array_equal(d['lr'], lr[i]) and np.array_equal(d['hr'], hr[i]) and d['lr'].shape == (32, 8, 8)
        assert d['cfa'] == [[2, 1], [3, 2]] and d['iso'] == 100 * (i + 1)          # GRBG + 1 (utils/sidd_utils.py:12)
        assert abs(d['reg'][0] - 1e-3 * (i + 1)) < 1e-12
        assert d['lr_path_full'].endswith('.npy') and np.array_equal(d['lr_full'], fulls[i])
        assert d['name'].startswith(f"{i:04d}_")
    x = dataload(str(tmp_path / "SIDD_Benchmark_Data" / "0000_000_GP_00100_00060_3200_L" / "0000_NOISY_RAW_010.MAT"))
    assert np.array_equal(x, fulls[0])                                             # v5 .mat with the variable 'x': transposed back
    with pytest.raises(RuntimeError):
        dataload("frame.dng")


def test_sidd_dataset_without_benchmark_dir(tmp_path):
    from yond_public_amd.data import SIDD_Dataset
    lr, hr, _ = make_tree(tmp_path, with_meta=False)
    d = SIDD_Dataset({'root_dir': str(tmp_path)})[1]
    assert d['cfa'] == [[1, 2], [2, 3]] and d['lr_path_full'] is None and d['lr_full'] is None


def _write_eld_tree(root, cams=("SonyA7S2",), scenes=(1, 2), H=32, W=48, bl=512, wp=16383):
    """A `.npy`-converted miniature of the ELD layout (<root>/<cam>/scene-<k>/IMG_<id>.npy next to where the ARW would be)."""
    import json
    rng = np.random.default_rng(3)
    for cam in cams:
        os.makedirs(root / cam, exist_ok=True)
        json.dump({"bl": bl, "wp": wp}, open(root / cam / "meta.json", "w"))
        for sc in scenes:
            d = root / cam / f"scene-{sc}"
            os.makedirs(d, exist_ok=True)
            for img in range(1, 17):
                np.save(d / f"IMG_{img:04d}.npy", rng.integers(bl - 20, wp, (H, W)).astype(np.uint16))


def test_eld_full_dataset_on_a_converted_tree(tmp_path):
    """data_process/yond_datasets.py:977-1067 on `.npy` copies: the file-id arithmetic (lr_id = 5 iso_id + ratio_id + 2, nearest
    long exposure of {1, 6, 11, 16}), change_eval_ratio's subset, the (raw - bl) * ratio / (wp - bl) scaling, the names."""
    from yond_public_amd.data import ELD_Full_Dataset
    root = tmp_path / "ELD"
    _write_eld_tree(root)
    ds = ELD_Full_Dataset({'root_dir': str(root), 'clip': False})
    assert len(ds) == 2 * 3 and ds.ratio == 1 and (ds.bl, ds.wp) == (512.0, 16383.0)        # 2 scenes x 3 ISOs at ratio 1
    ds.change_eval_ratio('SonyA7S2', ratio=100, iso_list=[1600])
    assert len(ds) == 2
    item = ds[0]
    # ISO 1600 -> iso_id 1, ratio 100 -> ratio_id 2: lr = IMG_0009, nearest long exposure IMG_0011
    assert item['name'] == 'SonyA7S2_01_IMG_0009' and item['ratio'] == 100 and item['ISO'] == 1600
    lr = np.load(root / "SonyA7S2" / "scene-1" / "IMG_0009.npy").astype(np.float32)
    hr = np.load(root / "SonyA7S2" / "scene-1" / "IMG_0011.npy").astype(np.float32)
    np.testing.assert_array_equal(item['lr'], ((lr - 512) * 100 / (16383 - 512)).astype(np.float32))
    np.testing.assert_array_equal(item['hr'], ((hr - 512) / (16383 - 512)).astype(np.float32))
    assert item['lr'].dtype == np.float32 and item['lr'].max() > 1.0                          # clip False: no clipping
    ds.change_eval_ratio('NikonD850', ratio=1)
    assert len(ds) == 0                                                                      # camera not converted: an empty subset
    # a raw file without its converted copy is a clear error, not a guess
    (root / "SonyA7S2" / "scene-1" / "IMG_0002.npy").unlink()
    (root / "SonyA7S2" / "scene-1" / "IMG_0002.ARW").write_bytes(b"")
    with pytest.raises(RuntimeError, match="rawpy"):
        ds.change_eval_ratio('SonyA7S2', ratio=1, iso_list=[800])
        ds[0]


def test_lrid_and_any_datasets(tmp_path, monkeypatch):
    """LRID (data_process/yond_datasets.py:870-975) on the converted tree -- evaluation scene ids, ratio switch, names -- and the
    plain directory of frames of the ANY runfile."""
    import json
    from yond_public_amd.data import LRID_Dataset, Any_Dataset
    monkeypatch.chdir(tmp_path)                                       # (no infos/*.info tables here: the tree is scanned)
    root = tmp_path / "LRID"
    rng = np.random.default_rng(4)
    for i in LRID_Dataset.get_eval_id('indoor_x5')[:3]:
        d = root / "indoor_x5" / f"{i:03d}"
        os.makedirs(d)
        np.save(d / "gt.npy", rng.integers(60, 1023, (24, 40)).astype(np.uint16))
        for r in (1, 2):
            np.save(d / f"x{r:02d}.npy", rng.integers(60, 600, (24, 40)).astype(np.uint16))
        json.dump({"wb": [2.0, 1.0, 1.0, 1.5], "ExposureTime": 0.01}, open(d / "meta.json", "w"))
    ds = LRID_Dataset({'root_dir': str(root), 'dstname': ['indoor_x5'], 'bl': 63, 'wp': 1023})
    assert len(ds) == 3 and (ds.H, ds.W) == (24, 40)
    ds.change_eval_ratio(2)
    it = ds[1]
    assert it['name'] == 'indoor_x5_014_x02' and it['ratio'] == 2 and it['ISO'] == 6400 and it['ExposureTime'] == 10.0
    raw = np.load(root / "indoor_x5" / "014" / "x02.npy").astype(np.float32)
    np.testing.assert_array_equal(it['lr'], ((raw - 63) * 2 / (1023 - 63)).astype(np.float32))
    np.testing.assert_allclose(it['wb'], [2.0, 1.0, 1.0, 1.5])
    frames = tmp_path / "frames"
    os.makedirs(frames / "gt")
    for k in range(2):
        np.save(frames / f"f{k}.npy", rng.integers(60, 700, (16, 24)).astype(np.uint16))
    np.save(frames / "gt" / "f1.npy", rng.integers(60, 700, (16, 24)).astype(np.uint16))
    a = Any_Dataset({'root_dir': str(frames), 'bl': 63, 'wp': 1023})
    a.change_eval_ratio(2)
    assert len(a) == 2 and 'hr' not in a[0] and 'hr' in a[1] and a[1]['name'] == 'f1_x02'
    np.testing.assert_array_equal(a[0]['lr'], ((np.load(frames / "f0.npy").astype(np.float32) - 63) * 2 / 960).astype(np.float32))


def test_greedy_shard_on_five_size_classes():
    """the SIDD validation set is 8 scenes from each of five phones with different frame sizes; the ranks get
    the images longest-first onto the least-loaded rank, from sizes known BEFORE anything is loaded (scene names / .npy headers)."""
    from yond_public_amd import data as Dt
    from yond_public_amd import distributed as D

    class Scenes:
        def __init__(self):
            phones = ['S6', 'GP', 'N6', 'G4', 'IP']
            self.infos = [{'name': f'{i:04d}_{i % 10:03d}_{phones[i % 5]}_00800_00350_3200_L', 'lr_path': None} for i in range(40)]

        def __len__(self):
            return len(self.infos)
    ds = Scenes()
    sizes = Dt.item_sizes(ds)
    assert len(sizes) == 40 and len(set(sizes)) == 5 and sizes[0] == 3000 * 5328
    world = 8
    shards = [D.shard_dataset(ds, r, world) for r in range(world)]
    assert sorted(i for s in shards for i in s) == list(range(40))                  # a partition
    loads = [sum(sizes[i] for i in s) for s in shards]
    rr = [sum(sizes[i] for i in range(r, 40, world)) for r in range(world)]         # round-robin's loads
    assert max(loads) <= max(rr) and max(loads) - min(loads) <= max(sizes)          # never worse, and within one frame of even
    assert max(loads) / (sum(loads) / world) <= 1.06
    # unknown sizes -> round-robin; one rank -> everything
    ds.infos[3]['name'] = 'unknown'
    assert Dt.item_sizes(ds) is None and D.shard_dataset(ds, 1, 8) == list(range(1, 40, 8))
    assert D.shard_dataset(ds, 0, 1) == list(range(40))


def test_npy_header_gives_the_item_size(tmp_path):
    from yond_public_amd import data as Dt

    class One:
        infos = []

        def __len__(self):
            return len(self.infos)
    p = str(tmp_path / "frame.npy")
    np.save(p, np.zeros((30, 52), np.float32))
    One.infos = [{'name': 'x', 'lr_path': p}]
    assert Dt.item_sizes(One()) == [30 * 52]


def test_prefetcher_orders_items_and_relays_errors():
    """The loader threads deliver the items in the order asked for, whatever order they finish in, as float32 tensors; an exception
    inside a worker surfaces at ITS item."""
    import time
    import torch
    from yond_public_amd.data import Prefetcher

    class Slow:
        def __len__(self):
            return 12

        def __getitem__(self, k):
            time.sleep(0.03 if k % 3 == 0 else 0.0)
            if k == 7:
                raise RuntimeError("item 7 is broken")
            return {'lr': np.full((2, 3), k, np.uint16), 'hr': None, 'name': f'i{k}'}
    got = []
    with pytest.raises(RuntimeError, match="item 7"):
        for k, d in Prefetcher(Slow(), [5, 0, 3, 9, 1, 7, 2], 'cpu', workers=3, depth=3):
            assert isinstance(d['lr'], torch.Tensor) and d['lr'].dtype == torch.float32 and float(d['lr'][0, 0]) == k and d['name'] == f'i{k}'
            got.append(k)
    assert got == [5, 0, 3, 9, 1]
    assert [k for k, _ in Prefetcher(Slow(), [0, 1, 2, 3], 'cpu', workers=8, depth=1)] == [0, 1, 2, 3]
    assert list(Prefetcher(Slow(), [], 'cpu')) == []


def test_prefetcher_look_ahead_is_gated_by_position():
    """No worker starts position p before the consumer has taken p - depth: with one slow item (position 1) the other workers may
    run at most `depth` positions past the consumer, and the slow position's worker is never starved of its turn."""
    import threading
    import time
    from yond_public_amd.data import Prefetcher
    started, lock = [], threading.Lock()

    class Items:
        def __len__(self):
            return 40

        def __getitem__(self, k):
            with lock:
                started.append(k)
            if k == 1:
                time.sleep(0.25)
            return {'lr': np.zeros((1, 1), np.float32), 'name': str(k)}
    pf = Prefetcher(Items(), list(range(40)), 'cpu', workers=4, depth=4)
    seen = []
    for k, _ in pf:
        with lock:
            ahead = max(started)
        assert ahead < len(seen) + 1 + pf.depth, (ahead, len(seen))
        seen.append(k)
    assert seen == list(range(40))
