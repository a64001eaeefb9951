#!/usr/bin/env python3
"""Only the 256 x 256 fixture of make_golden.py (the reference imported from /root/reference, same shims): BASELINE configs[0]'s size.
    python tests/golden/make_golden_256.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg    # noqa: E402

if __name__ == '__main__':
    mg.install_shims()
    torch.set_num_threads(8)
    mg.gen_e2e('train_256x256_b1', 1, 256, 256, True, 'bern', stages=False, compact=True)
