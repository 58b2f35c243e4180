"""`from yond_public_amd.archs import *` mirrors `from archs import *` (YOND_SIDD.py:7) for the
hot-path denoisers; classes are resolved by name from the runfile's arch['name']."""
from .unet import GuidedResUnet, SNRnet, UNetSeeInDark

__all__ = ["GuidedResUnet", "SNRnet", "UNetSeeInDark"]
