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
    return pkg.hipbind.MiganLib(build(), allow_test_backend=True)


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


# ---- 16-bit activation storage (MIGAN_DTYPE_BF16 / MIGAN_DTYPE_F16): tensors as the library stores them ----------------
def to_storage(a, storage):
    """fp32 array -> the array the library reads/writes in `storage` (fp32, or uint16 bit patterns of bf16 / fp16), aligned"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if storage in (None, "f32"):
        return aligned(a)
    if storage == "bf16":
        u = a.view(np.uint32)
        bits = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
        nan = np.isnan(a)
        bits[nan] = 0x7FC0
    else:
        bits = a.astype(np.float16).view(np.uint16)
    buf = np.empty(bits.size + 8, dtype=np.uint16)
    off = (16 - buf.ctypes.data % 16) % 16 // 2
    out = buf[off:off + bits.size].reshape(bits.shape)
    out[...] = bits
    return out


def from_storage(a, storage):
    if storage in (None, "f32"):
        return np.asarray(a, dtype=np.float32)
    a = np.ascontiguousarray(a)
    if storage == "bf16":
        return (a.astype(np.uint32) << np.uint32(16)).view(np.float32)
    return a.view(np.float16).astype(np.float32)


def storage_ulp(ref, storage):
    """spacing of the storage format at |ref| (fp32: of float32)"""
    bits = {"bf16": 7, "f16": 10}.get(storage, 23)
    mag = np.maximum(np.abs(ref).astype(np.float64), 2.0 ** -14 if storage == "f16" else 2.0 ** -126)
    return np.exp2(np.floor(np.log2(mag)) - bits)


def storage_close(got, want, storage, ulps=1, frac=0.02, what="", top_ulps=0.0):
    """`got` (read back from `storage`) equals `want` (the oracle rounded to `storage`) except where the two fp32 values
    differed in their last places just before rounding: those elements may land `ulps` storage steps apart, and there
    must be few of them.  `top_ulps`: additional absolute slack in storage steps of the tensor's LARGEST magnitude, for
    tensors several layers deep (an upstream element that rounded the other way perturbs everything it feeds by its own
    step size times a weight, whatever the magnitude of the element it lands on)."""
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    if storage in (None, "f32"):
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want).max())), err_msg=what)
        return
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    slack = ulps * storage_ulp(want, storage) + 2e-5 * max(1.0, float(np.abs(want).max()))
    slack = slack + top_ulps * float(storage_ulp(np.abs(want).max(), storage))
    bad = diff > slack
    assert not bad.any(), f"{what}: {int(bad.sum())} elements more than {ulps} {storage} steps off, worst {float(diff.max()):.3e}"
    moved = float((diff > 2e-5 * max(1.0, float(np.abs(want).max()))).mean())
    assert moved <= frac, f"{what}: {moved:.3%} of the elements rounded differently (allowed {frac:.1%})"
