"""Helpers shared by the CPU tests that drive the product's C ABI through the fiber emulator
(tests/emu).  Test infrastructure only."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def emu_lib():
    from tests.emu.build_emu import build
    pkg = importlib.import_module("mi-gan_amd")
    return pkg.hipbind.MiganLib(build())


def nhwc(a):
    return np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))


def nchw(a):
    return np.ascontiguousarray(np.transpose(a, (0, 3, 1, 2)))


def ptr(a):
    return None if a is None else a.ctypes.data


def aligned(a):
    """contiguous float32 copy whose data pointer is 16-byte aligned"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ctypes.data % 16 == 0:
        return a
    buf = np.empty(a.size + 4, dtype=np.float32)
    off = (16 - buf.ctypes.data % 16) % 16 // 4
    out = buf[off:off + a.size].reshape(a.shape)
    out[...] = a
    return out
