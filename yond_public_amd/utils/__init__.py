"""`from yond_public_amd.utils import *` mirrors the hot-path names of `from utils import *` (YOND_SIDD.py:6)."""
from .isp_ops import bayer2rggb, rggb2bayer, bayer2rggbs, rggb2bayers
from .isp_algos import VST, inverse_VST, get_bias, stdfilt, varfilt, polyfit
from .sidd_utils import rot_bayer, read_metadata
from ..data import dataload
from ..pipeline import get_p2d, SimpleNLF, get_threshold, VST_Denoiser, Simple_Denoiser, IterDenoise

__all__ = ["bayer2rggb", "rggb2bayer", "bayer2rggbs", "rggb2bayers", "VST", "inverse_VST", "get_bias", "stdfilt", "varfilt",
           "polyfit", "rot_bayer", "read_metadata", "dataload", "get_p2d", "SimpleNLF", "get_threshold", "VST_Denoiser", "Simple_Denoiser", "IterDenoise"]
