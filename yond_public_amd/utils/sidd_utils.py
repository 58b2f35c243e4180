"""Function seam of utils/sidd_utils.py for the path: rot_bayer (:198-213) and the SIDD metadata reader (:3-73)."""
import numpy as np
import torch

from .. import _lib as L
from .. import pipeline as _P

_ROT_K = {((1, 2), (2, 3)): 0, ((2, 1), (3, 2)): 3, ((2, 3), (1, 2)): 1, ((3, 2), (2, 1)): 2}     # utils/sidd_utils.py:199-206


def rot_k(bayer_pattern, rev=False):
    """Number of counter-clockwise quarter turns that bring `bayer_pattern` (2x2, 1=R 2=G 3=B) to RGGB (:199-211)."""
    key = tuple(tuple(int(v) for v in row) for row in np.asarray(bayer_pattern).reshape(2, 2).tolist())
    if key not in _ROT_K:
        raise ValueError(f"Unknown Bayer pattern {bayer_pattern!r}")
    k = _ROT_K[key]
    return (4 - k) % 4 if rev else k


def rot_bayer(image, bayer_pattern, rev=False, axis=(-2, -1)):
    """utils/sidd_utils.py:198-213: np.rot90 by the pattern's quarter turns.  NumPy in -> NumPy out (host index
    permutation, as the reference); a device tensor [.., H, W] is rotated by the HIP copy kernel."""
    k = rot_k(bayer_pattern, rev)
    if isinstance(image, np.ndarray):
        return np.rot90(image, k=k, axes=axis)
    if tuple(a % image.dim() for a in axis) != (image.dim() - 2, image.dim() - 1):
        raise L.YondHipError("device rot_bayer rotates the last two axes")
    return _P.rot90(image, k)


def read_metadata(metadata):
    """utils/sidd_utils.py:3-21 for a scipy.io.loadmat'ed SIDD METADATA_RAW .MAT: noise model, CFA, white balance, colour matrix."""
    meta = metadata['metadata'][0, 0]
    beta1, beta2 = meta['UnknownTags'][7, 0][2][0][0:2]
    model = meta['Make'][0]
    cam = {'Apple': 'IP', 'Google': 'GP', 'samsung': 'S6', 'motorola': 'N6', 'LGE': 'G4'}[model]
    bayer_pattern = _get_bayer_pattern(meta)
    if cam == 'S6':                                     # :9-11: "the correct Bayer pattern is GBRG in S6"
        bayer_pattern = [1, 2, 0, 1]
    bayer_2by2 = (np.asarray(bayer_pattern) + 1).reshape((2, 2)).tolist()
    try:
        iso = meta['ISOSpeedRatings'][0][0]
    except Exception:
        iso = meta['DigitalCamera'][0, 0]['ISOSpeedRatings'][0][0]
    return {'meta': meta, 'beta1': beta1, 'beta2': beta2, 'bayer_2by2': bayer_2by2, 'wb': meta['AsShotNeutral'],
            'cst2': meta['ColorMatrix2'].reshape((3, 3)), 'iso': iso, 'cam': cam}


def _get_bayer_pattern(meta):
    """utils/sidd_utils.py:40-68: tag 33422 in UnknownTags / SubIFDs; RGGB when absent."""
    bayer_id, idx = 33422, 1
    for get in (lambda: meta['UnknownTags'], lambda: meta['SubIFDs'][0, 0]['UnknownTags'][0, 0], lambda: meta['SubIFDs'][0, 1]['UnknownTags']):
        try:
            tags = get()
            if tags[idx]['ID'][0][0][0] == bayer_id:
                return tags[idx]['Value'][0][0]
        except Exception:
            continue
    return [1, 2, 2, 3]
