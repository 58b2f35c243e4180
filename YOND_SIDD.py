#!/usr/bin/env python
"""Entry-point shim: `python YOND_SIDD.py -f runfiles/YOND/SIDD_simple+full_pre_grumix.yml -m eval` (reference CLI,
YOND_SIDD.py:723-744) -> yond_public_amd.YOND_SIDD."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from yond_public_amd.YOND_SIDD import main  # noqa: E402

if __name__ == '__main__':
    main()
