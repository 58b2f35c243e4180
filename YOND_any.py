#!/usr/bin/env python
"""Entry-point shim (README.md:38-47 names YOND_any.py; the reference does not ship it): the full-frame driver
yond_public_amd.YOND_full with the reference's runfile schema (runfiles/YOND/*_simple+full_pre_grumix.yml)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from yond_public_amd.YOND_full import main  # noqa: E402

if __name__ == '__main__':
    main()
