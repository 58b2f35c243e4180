"""CPU: the est_types that read round 1's estimate from files of the dataset tree (YOND_SIDD.py:316-337) -- the lookups themselves
(yond_public_amd.pipeline.file_estimate: host-side file reads, no GPU)."""
import pickle

import numpy as np
import pytest
import scipy.io as sio


def test_file_estimate_lookups(tmp_path):
    from yond_public_amd import pipeline as P
    from yond_public_amd._lib import YondHipError
    raw = tmp_path / 'SIDD_Validation_Raw'
    raw.mkdir()
    rng = np.random.default_rng(0)
    foi, liu, zou, pge = (rng.random((6, 2)) * 1e-2 for _ in range(4))
    sio.savemat(str(raw / 'FoiEst_fullPict.mat'), {'return_params': foi})
    sio.savemat(str(raw / 'LiuEst_fullPict.mat'), {'return_params': liu})
    np.save(str(raw / 'Zou_fullPict.npy'), zou)
    np.save(str(raw / 'PGE_fullPict.npy'), pge)
    est = {'root_dir': str(tmp_path), 'img_id': 4, 'name': '0009_001_S6_00800_00350_3200_L'}
    for t, tab in (('foi', foi), ('liu', liu), ('zou', zou), ('foi+full', foi)):
        assert tuple(P.file_estimate({'est_type': t}, est)) == (tab[4, 0], tab[4, 1])                 # :324-329
    r = P.file_estimate({'est_type': 'pge'}, est)
    assert r[0] == pge[4, 0] and r[1] == pge[4, 1] ** 2                                               # :337
    assert P.file_estimate({'est_type': 'simple'}, est) is None and P.file_estimate({'est_type': 'manual'}, None) is None
    with pytest.raises(YondHipError):
        P.file_estimate({'est_type': 'zou'}, {'img_id': 1})                                           # no dataset directory
    with pytest.raises(NotImplementedError):
        P.file_estimate({'est_type': 'pge'}, dict(est, est_net=object()))                             # :333-335
    # the calibration record (:316-323): a stored (camera, ISO) pair, else the camera's polynomials in the ISO
    cal = tmp_path / 'cal.pkl'
    rec = {'sfrn': {'S6_00800': (1.5e-3, 2.5e-6)}, 'beta1': {'S6': [2e-6, 1e-4]}, 'beta2': {'S6': [1e-9, 0.0, 3e-7]}}
    with open(cal, 'wb') as f:
        pickle.dump(rec, f)
    assert tuple(P.file_estimate({'est_type': 'simple', 'cal_est': str(cal)}, est)) == (1.5e-3, 2.5e-6)
    r = P.file_estimate({'est_type': 'foi', 'cal_est': str(cal)}, dict(est, name='0009_001_S6_00100_00350_3200_L'))   # cal_est comes first
    assert r[0] == np.poly1d(rec['beta1']['S6'])(100) and r[1] == np.poly1d(rec['beta2']['S6'])(100)
