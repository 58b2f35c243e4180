"""CPU: the product's synthetic generators are bit-identical to the ones the fixtures were made with."""
import numpy as np
import torch

import yond_oracle as O


def test_frames_identical():
    from yond_public_amd import synthetic as S
    a, ca = S.synth_noisy(64, 96, 4.0, 6.0, 3)
    b, cb = O.synth_noisy(64, 96, 4.0, 6.0, 3)
    assert np.array_equal(a, b) and np.array_equal(ca, cb)


def test_weights_identical():
    from yond_public_amd import synthetic as S
    from yond_public_amd.archs import GuidedResUnet, SNRnet, UNetSeeInDark
    for cls in (GuidedResUnet, SNRnet, UNetSeeInDark):
        arch = dict(name=cls.__name__, in_nc=4, out_nc=4, nf=8, nframes=1, res=True, norm=True)
        mine = S.procedural_state_dict(cls(arch), 4)
        ref = O.procedural_state_dict(arch, 4)
        assert mine.keys() == ref.keys()
        for k in ref:
            assert torch.equal(mine[k], ref[k]), k
