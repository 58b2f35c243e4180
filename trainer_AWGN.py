#!/usr/bin/env python
"""Entry-point shim: `python trainer_AWGN.py -f runfiles/Gaussian/GRU_5to50_norm_mix.yml -m train` (reference CLI,
trainer_AWGN.py:366-404) -> yond_public_amd.trainer_AWGN."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from yond_public_amd.trainer_AWGN import main  # noqa: E402

if __name__ == '__main__':
    main()
