"""`from yond_public_amd.archs import *` mirrors `from archs import *` (YOND_SIDD.py:7) for the
hot-path denoisers; classes are resolved by name from the runfile's arch['name']."""
import torch.nn as nn

from .unet import GuidedResUnet, SNRnet, UNetSeeInDark


def initialize_weights(net):
    """archs/__init__.py:10-17: N(0, 0.02) on every convolution's weight and bias and on every transposed convolution's weight."""
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data.normal_(0.0, 0.02)
            if m.bias is not None:
                m.bias.data.normal_(0.0, 0.02)
        if isinstance(m, nn.ConvTranspose2d):
            m.weight.data.normal_(0.0, 0.02)


__all__ = ["GuidedResUnet", "SNRnet", "UNetSeeInDark", "initialize_weights"]
