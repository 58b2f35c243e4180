#!/usr/bin/env python
"""Entry-point shim (README.md:38-47 names YOND_DND.py; the reference does not ship it).  The reference's DND runfile
(runfiles/YOND/DND_simple+full_pre_grumix.yml) is the SIDD dataset stack with `full_dn: True` and no code reads its
`data_type`, so this is yond_public_amd.YOND_SIDD with that runfile as the default (`-m eval` for the validation blocks)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from yond_public_amd.YOND_SIDD import main  # noqa: E402

if __name__ == '__main__':
    argv = sys.argv[1:]
    if '-f' not in argv and '--runfile' not in argv:
        argv = ['-f', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'runfiles', 'YOND', 'DND_simple+full_pre_grumix.yml')] + argv
    main(argv)
