#!/usr/bin/env python3
"""Measurement build of libmigan_hip.so under another name: python scripts/build_variant.py <name> [-DFLAG ...]
-> mi-gan_amd/csrc/libmigan_hip_<name>.so (objects in mi-gan_amd/csrc/_obj_<name>/), selected at run time with
MIGAN_HIP_LIBRARY=<that path>.  Never the product: build.py's stamp only vouches for libmigan_hip.so."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("mi-gan_amd.build")
print(b.build_variant(sys.argv[1], sys.argv[2:], force=True))
