#!/usr/bin/env python3
"""Measurement build of libmigan_hip.so under another name: python scripts/build_variant.py <name> [-DFLAG ...]
-> mi-gan_amd/csrc/libmigan_hip_<name>.so (objects in mi-gan_amd/csrc/_obj_<name>/), selected at run time with
MIGAN_HIP_LIBRARY=<that path>.  Never the product: build.py's stamp only vouches for libmigan_hip.so."""
import importlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("mi-gan_amd.build")
name, extra = sys.argv[1], sys.argv[2:]
obj = os.path.join(b.CSRC, f"_obj_{name}")
os.makedirs(obj, exist_ok=True)
out = os.path.join(b.CSRC, f"libmigan_hip_{name}.so")
units = []
for o, cmd in b.units(extra):
    o2 = os.path.join(obj, os.path.basename(o))
    units.append((o2, cmd[:-1] + [o2]))


def run(u):
    r = subprocess.run(u[1], cwd=b.CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    return u[0]


with ThreadPoolExecutor(max_workers=os.cpu_count()) as ex:
    objs = list(ex.map(run, units))
subprocess.run([b.hipcc(), f"--offload-arch={b.ARCH}", "-shared", "-fPIC", *objs, "-o", out], check=True, cwd=b.CSRC)
print(out)
