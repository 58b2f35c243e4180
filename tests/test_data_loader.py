"""CPU: dataset ingestion (N3) -- dataload and SIDD_Dataset on a miniature tree in the reference's layout
(data_process/yond_datasets.py:767-868, utils/utils.py:319-335), written here with scipy.io.savemat (MATLAB v5)."""
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
            sio.savemat(d / f"{i:04d}_NOISY_RAW_010.MAT", {"x": full})          # stands for the v7.3 file; the .npy copy is preferred
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
    assert np.array_equal(x, fulls[0])                                             # v5 .mat with the variable 'x'
    with pytest.raises(RuntimeError):
        dataload("frame.dng")


def test_sidd_dataset_without_benchmark_dir(tmp_path):
    from yond_public_amd.data import SIDD_Dataset
    lr, hr, _ = make_tree(tmp_path, with_meta=False)
    d = SIDD_Dataset({'root_dir': str(tmp_path)})[1]
    assert d['cfa'] == [[1, 2], [2, 3]] and d['lr_path_full'] is None and d['lr_full'] is None
