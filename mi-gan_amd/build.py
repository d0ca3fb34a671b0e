"""Ahead-of-time build of libmigan_hip.so for gfx950 (MI355X) with hipcc.

In-tree on purpose: the .so sits next to its sources (mi-gan_amd/csrc/) so it travels with the
repository snapshot to the GPU box and shows up as a loaded in-tree native library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OUT = os.path.join(CSRC, "libmigan_hip.so")
SOURCES = [os.path.join(CSRC, f) for f in ("migan_hip.hip", "migan_kernels.hpp", "migan_host.hpp", "migan_rt_hip.h",
                                            "comodgan_kernels.hpp", "comodgan_host.hpp")] + [
    os.path.join(ROOT, "include", "migan_hip.h"), os.path.join(ROOT, "include", "comodgan_hip.h")]
ARCH = "gfx950"


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for --offload-arch=gfx950)")


def command(extra: List[str] = ()) -> List[str]:
    return [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-honor-nans",
            os.path.join(CSRC, "migan_hip.hip"), "-o", OUT, *extra]


def is_fresh() -> bool:
    if not os.path.exists(OUT):
        return False
    return os.path.getmtime(OUT) >= max(os.path.getmtime(s) for s in SOURCES)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and is_fresh():
        return OUT
    cmd = command()
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
