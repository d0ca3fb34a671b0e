#!/usr/bin/env python
"""Whole-generator error of the split-operand GEMM schemes, simulated on the numpy oracle
(test infrastructure, CPU only).  The 1x1 feature convolutions are replaced by sums of matrix
products of low-precision pieces (each product exact in fp32, as in the MFMA accumulator);
everything else stays fp32.  Reference = the same network in float64.

  python scripts/split_accuracy.py [RES] [SEED]
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import migan_oracle as orc  # noqa: E402

pkg = importlib.import_module("mi-gan_amd")


def to_bf16(x):  # RNE
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def to_f16(x, rtz=False):
    x = x.astype(np.float32)
    h = x.astype(np.float16).astype(np.float32)
    if rtz:
        over = np.abs(h) > np.abs(x)
        h16 = x.astype(np.float16)
        h = np.where(over, np.nextafter(h16, np.float16(0)).astype(np.float32), h)
    return h


def split(x, n, conv):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        h = conv(r)
        out.append(h)
        r = (r - h).astype(np.float32)
    return out


def pow2_scale(maxabs, target_exp):
    e = np.floor(np.log2(max(float(maxabs), 1e-30)))
    return np.float32(2.0 ** (target_exp - e))


def make_pointwise(scheme):
    exact = orc.pointwise

    def pw(x, w, bias=None):
        n, c, h, wd = x.shape
        co = w.shape[0]
        if x.dtype != np.float32 or c < 32 or co < 32 or bias is not None:
            return exact(x, w, bias)
        A = x.reshape(n, c, h * wd)
        W = w.reshape(co, c).astype(np.float32)
        if scheme == "f32":
            return exact(x, w, bias)
        if scheme.startswith("bf16x3"):
            a, b = split(A, 3, to_bf16), split(W, 3, to_bf16)
            terms = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)] if scheme == "bf16x3_6" else [(1, 0), (0, 1), (0, 0)]
            inv = np.float32(1.0)
        else:
            rtz = "rtz" in scheme
            sa = np.float32(2.0 ** 7)                     # |a| <= 256 (lrelu_agc clamp) -> < 2^15
            sw = pow2_scale(np.abs(W).max(), 13)          # max|w| scaled into [2^13, 2^14)
            conv = lambda v: to_f16(v, rtz)
            a, b = split(A * sa, 2, conv), split(W * sw, 2, conv)
            terms = [(1, 1), (1, 0), (0, 1), (0, 0)] if "_4" in scheme else [(1, 0), (0, 1), (0, 0)]
            inv = np.float32(1.0) / (sa * sw)
        acc = np.zeros((n, co, h * wd), np.float32)
        for i, j in terms:                                 # smallest products first
            acc += np.matmul(b[j], a[i])
        return (acc * inv).reshape(n, co, h, wd)
    return pw


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    sd = pkg.synth.make_state_dict(res, seed=seed)
    x = pkg.synth.make_input(1, res, seed=seed)
    ref = orc.generator(x, sd, res, dtype=np.float64)
    print(f"res {res} seed {seed} |y|max {np.abs(ref).max():.2f}")
    exact = orc.pointwise
    for scheme in ("f32", "bf16x3_6", "bf16x3_3", "f16x2_3", "f16x2_3_rtz", "f16x2_4", "f16x2_4_rtz"):
        orc.pointwise = make_pointwise(scheme)
        try:
            y = orc.generator(x, sd, res)
        finally:
            orc.pointwise = exact
        d = np.abs(y.astype(np.float64) - ref)
        print(f"{scheme:14s} max abs err vs fp64 {d.max():.3e}   rms {np.sqrt((d ** 2).mean()):.3e}", flush=True)


if __name__ == "__main__":
    main()
