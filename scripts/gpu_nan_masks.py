#!/usr/bin/env python3
"""NaN masks of a library build against the oracle: one NaN input element per operator case (tests/sepconv_case.py nan_at), and a
generator forward with one NaN pixel against the torch-CPU port of the reference.  usage: gpu_nan_masks.py <libmigan_hip*.so>"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mi-gan_amd")
from tests.sepconv_case import CudaMem, run_sepconv_case
lib = pkg.hipbind.MiganLib(sys.argv[1])
print("library", sys.argv[1], "policy", lib.nan_policy())
mem = CudaMem(torch.device("cuda", 0))
cases = [dict(cin=64, cout=64, h=32, batch=2), dict(cin=64, cout=128, h=16, batch=2, down=2),
         dict(cin=128, cout=64, h=16, batch=2, up=2, noise=True, skip=True), dict(cin=256, cout=256, h=16, batch=1),
         dict(cin=64, cout=64, h=32, batch=2, fromrgb=True), dict(cin=64, cout=64, h=16, batch=2, noise=True, torgb=True, with_prev=True),
         dict(cin=256, cout=256, h=32, batch=4, noise=True), dict(cin=512, cout=512, h=32, batch=2), dict(cin=64, cout=128, h=64, batch=4, down=2),
         dict(cin=128, cout=128, h=32, batch=2, noise=True, torgb=True, with_prev=True)]
for min_tiles in (256, 1):
    lib.set_tuning("pipe_min_tiles", min_tiles)
    lib.set_tuning("w2_min_tiles", min_tiles)
    for kw in cases:
        nan_at = (0, 2, 7, 9) if kw.get("fromrgb") else (0, 5, 7, 9)
        run_sepconv_case(lib, pkg, mem, seed=13, nan_at=nan_at, **kw)
        print("ok", min_tiles, kw, lib.last_kernel())
print("ALL MASKS OK")
