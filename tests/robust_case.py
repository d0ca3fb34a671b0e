"""Adversarial INTRA-tensor operand ranges for the split GEMM: the f16x2 variant multiplies fp16 hi / lo
pieces of operands scaled by one power of two per tensor (activations x 2^7, weights so that max|w| lands in [2^13, 2^14)); a tensor
whose elements span many orders of magnitude pushes the lo pieces of its small elements into fp16's subnormal range (kept, not
flushed, by v_cvt_pk_f16_f32 and v_mfma_f32_32x32x16_f16: profiles/r03_ubench_mfma_fp16_subnormals.txt) or below it.  Each case runs
one SeparableConv2d through migan_sepconv_forward with the exact-fp32 MFMA and with f16x2 and measures both against a float64
evaluation of the same layer.  Shared by the emulator (CPU) and the GPU tests.  Test infrastructure only."""
import numpy as np

from oracle import migan_oracle as orc
from tests.emu_util import aligned, nchw, nhwc

GEMM_F32, GEMM_F16X2 = 0, 2


def make_case(pkg, kind, cin=64, cout=64, h=16, batch=1, seed=11):
    s = pkg.synth
    w1 = np.zeros((cin, 1, 3, 3), np.float32)
    w1[:, 0, 1, 1] = 1.0                                     # identity depthwise: the GEMM's A operand is act(x), chosen below
    b1 = np.zeros((cin,), np.float32)
    x = (s.normal((batch, cin, h, h), seed, "x") * 1.5).astype(np.float32)
    w2 = (s.normal((cout, cin, 1, 1), seed, "w2") / np.sqrt(cin)).astype(np.float32)
    if kind == "weight_outlier":
        # one weight per output row 1e4 x the others: the per-tensor scale is set by the outliers, every other weight's hi piece
        # has a few significant bits above fp16's subnormal threshold and its lo piece is subnormal
        for co in range(cout):
            w2[co, (7 * co + 3) % cin, 0, 0] *= np.float32(1e4)
    elif kind == "activation_mix":
        # K rows mixing clamp-saturated (+-256) and 1e-4-sized activations
        x = (s.normal((batch, cin, h, h), seed, "x") * 1e-4).astype(np.float32)
        x[:, ::3] = np.sign(x[:, ::3]) * np.float32(1e3)      # lrelu_agc -> +256 / -256 (clamp)
    elif kind == "lo_subnormal":
        # weights spread over 2^-20 .. 1 of the tensor maximum: hi pieces down to fp16 subnormals, lo pieces below them
        e = (np.arange(cout * cin).reshape(cout, cin, 1, 1) * 7) % 21
        w2 = (np.sign(w2) * np.exp2(-e.astype(np.float32)) * (1.0 + 0.37 * np.abs(w2))).astype(np.float32)
    elif kind == "tiny_everything":
        w2 *= np.float32(1.7e-5)
        x *= np.float32(1e-3)
    else:
        raise ValueError(kind)
    return x, dict({"m.conv1.weight": w1, "m.conv1.bias": b1, "m.conv2.weight": w2})


def run_case(lib, pkg, mem, kind, gemm, **kw):
    """(max |err| against float64, max |float64 result|) of one SeparableConv2d run with GEMM variant `gemm`"""
    x, sd = make_case(pkg, kind, **kw)
    batch, cin, h, _ = x.shape
    cout = sd["m.conv2.weight"].shape[0]
    want = orc.separable_conv(x.astype(np.float64), {k: v.astype(np.float64) for k, v in sd.items()}, "m")
    keep = []

    def dev(a):
        keep.append(mem.put(aligned(a)))
        return keep[-1]

    xin = dev(nhwc(x))
    y = dev(np.full((batch, h, h, cout), np.nan, np.float32))
    w1, b1, w2 = dev(sd["m.conv1.weight"]), dev(sd["m.conv1.bias"]), dev(sd["m.conv2.weight"])
    wsp_n = (3 * cin * cout + 1) // 2 + 8
    wsp = dev(np.full(wsp_n, np.nan, np.float32))
    lib.sepconv_forward(stream=mem.stream, x=mem.ptr(xin), y=mem.ptr(y), skip=None, conv1_weight=mem.ptr(w1), conv1_bias=mem.ptr(b1),
                        conv2_weight=mem.ptr(w2), noise_const=None, noise_strength=None, batch=batch, cin=cin, cout=cout, res_in=h, width_in=h,
                        down=1, up=1, scratch=None, scratch_bytes=0, wsplit=mem.ptr(wsp), wsplit_bytes=wsp_n * 4, dtype=0, gemm=gemm)
    mem.sync()
    got = nchw(mem.get(y)).astype(np.float64)
    assert np.isfinite(got).all()
    return float(np.abs(got - want).max()), float(np.abs(want).max())


KINDS = ("weight_outlier", "activation_mix", "lo_subnormal", "tiny_everything")


def check_kind(lib, pkg, mem, kind, **kw):
    """f16x2 must be as good as the exact-fp32 MFMA path on the same operands: err(f16x2) <= 2 err(f32) + 1e-6 |y|max"""
    e32, ymax = run_case(lib, pkg, mem, kind, GEMM_F32, **kw)
    e16, _ = run_case(lib, pkg, mem, kind, GEMM_F16X2, **kw)
    assert e32 <= 2e-6 * max(ymax, 1e-30) * 64, (kind, e32, ymax)      # the fp32 path itself: K = 64 products of fp32 rounding
    assert e16 <= 2.0 * e32 + 1e-6 * ymax, f"{kind}: f16x2 error {e16:.3e} vs exact-fp32 error {e32:.3e} at |y|max {ymax:.3e}"
    return e32, e16, ymax
