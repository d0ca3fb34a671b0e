"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement in numpy of the MI-GAN inference generator forward,
``lib/model_zoo/migan_inference.py`` of Picsart-AI-Research/MI-GAN.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.

Pinning: the reference has no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is therefore pinned against outputs of the
reference module itself, produced in the build container by
``tests/golden/make_golden.py`` (imports /root/reference) and committed under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
below against them.

All tensors are NCHW numpy arrays like the reference's torch tensors.  ``dtype``
selects float32 (default, the reference's precision) or float64 (a
higher-precision truth for error budgeting).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

SQRT2_F32 = np.float32(np.sqrt(2))        # reference :15, applied as an fp32 multiply :25


# --------------------------------------------------------------------------- 16-bit activation storage (BASELINE configs[1])
def round_storage(x: np.ndarray, storage: Optional[str]) -> np.ndarray:
    """What a tensor reads back as after being stored in `storage` ('f32' / None: unchanged; 'bf16', 'f16': round to
    nearest even).  The 16-bit modes of the HIP library (MIGAN_DTYPE_BF16 / _F16, include/migan_hip.h) keep parameters,
    network input/output, the running RGB image and all arithmetic of the reference (:154-170) in fp32 and round once
    per stored feature map: every SeparableConv2d output, after the skip add where the block has one (:272, :305)."""
    if storage in (None, "f32"):
        return x
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    if storage == "bf16":
        u = x32.view(np.uint32)
        r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
        out = r.view(np.float32)
    elif storage == "f16":
        out = x32.astype(np.float16).astype(np.float32)
    else:
        raise ValueError(storage)
    return out.astype(x.dtype)


def round_gemm_operands(x: np.ndarray, w: np.ndarray):
    """GEMM variant "f16" of the 16-bit storage modes (MIGAN_GEMM_F16): both operands of the 1x1 convolution are rounded to fp16
    (11-bit significands, round to nearest even) after exact power-of-two scaling -- activations by 2^7 (they are bounded by the
    +-256 clamp), each weight tensor so that its largest magnitude lands in [2^13, 2^14) -- products and sums are fp32."""
    x32 = np.asarray(x, dtype=np.float32)
    xr = ((x32 * np.float32(128.0)).astype(np.float16).astype(np.float32) / np.float32(128.0)).astype(x.dtype)
    w32 = np.asarray(w, dtype=np.float32)
    m = float(np.abs(w32).max())
    if m > 0 and np.isfinite(m):
        e = int(np.floor(np.log2(m)))
        e = min(max(e, -100), 100)
        s = np.float32(2.0 ** (13 - e))
        wr = ((w32 * s).astype(np.float16).astype(np.float32) / s)
    else:
        wr = w32
    return xr, wr.astype(w.dtype)


def noise_plane(noise_const: np.ndarray, h: int, w: int) -> np.ndarray:
    """noise_const [r,r] at an h x w layer (arbitrary-size forward, reference README.md:87): tiled periodically and
    cropped; the identity at h = w = r."""
    r0, r1 = noise_const.shape
    if (h, w) == (r0, r1):
        return noise_const
    return np.tile(noise_const, ((h + r0 - 1) // r0, (w + r1 - 1) // r1))[:h, :w]


# --------------------------------------------------------------------------- a1
def lrelu_agc(x: np.ndarray, alpha: float = 0.2, clamp: float = 256.0) -> np.ndarray:
    """leaky_relu(x, 0.2) * sqrt(2), clamped to +-256  (reference :20-28)."""
    dt = x.dtype
    y = np.where(x > 0, x, x * dt.type(alpha))
    gain = SQRT2_F32.astype(dt) if dt == np.float32 else dt.type(np.sqrt(2))
    y = y * gain
    return np.clip(y, dt.type(-clamp), dt.type(clamp))


# --------------------------------------------------------------------------- a2
def fir_taps(gain: float) -> np.ndarray:
    """setup_filter([1,3,3,1], gain) (reference :31-55): outer(f,f)/64*gain."""
    f = np.array([1.0, 3.0, 3.0, 1.0])
    k = np.outer(f, f)
    k = k / k.sum()
    return k * (gain ** (k.ndim / 2))


def _depthwise(x: np.ndarray, w: np.ndarray, pad: Tuple[int, int, int, int], stride: int = 1,
               bias: Optional[np.ndarray] = None) -> np.ndarray:
    """Cross-correlation with one KxK filter per channel (torch Conv2d, groups=C).
    x [N,C,H,W], w [C,1,K,K], pad = (top, bottom, left, right)."""
    n, c, h, wd = x.shape
    k = w.shape[-1]
    xp = np.pad(x, ((0, 0), (0, 0), (pad[0], pad[1]), (pad[2], pad[3])))
    ho = (h + pad[0] + pad[1] - k) // stride + 1
    wo = (wd + pad[2] + pad[3] - k) // stride + 1
    out = np.zeros((n, c, ho, wo), dtype=x.dtype)
    if bias is not None:
        out += bias.reshape(1, c, 1, 1).astype(x.dtype)
    for ky in range(k):
        for kx in range(k):
            tap = w[:, 0, ky, kx].reshape(1, c, 1, 1).astype(x.dtype)
            out += tap * xp[:, :, ky:ky + stride * ho:stride, kx:kx + stride * wo:stride]
    return out


# --------------------------------------------------------------------------- a3
def downsample2d(x: np.ndarray, filt: Optional[np.ndarray] = None) -> np.ndarray:
    """Downsample2d (reference :58-76): depthwise 4x4, stride 2, zero pad 1,
    taps [1,3,3,1] (x) [1,3,3,1] / 64."""
    c = x.shape[1]
    if filt is None:
        filt = np.broadcast_to(fir_taps(1.0), (c, 1, 4, 4))
    return _depthwise(x, filt.astype(x.dtype), (1, 1, 1, 1), stride=2)


# --------------------------------------------------------------------------- a4
def upsample2d(x: np.ndarray, filt: Optional[np.ndarray] = None) -> np.ndarray:
    """Upsample2d (reference :79-103): nearest x2, multiply by filter_const
    (keeps even/even samples only = zero insertion), pad (2,1,2,1), depthwise
    4x4 with taps outer/16."""
    n, c, h, w = x.shape
    if filt is None:
        filt = np.broadcast_to(fir_taps(4.0), (c, 1, 4, 4))
    z = np.zeros((n, c, 2 * h, 2 * w), dtype=x.dtype)
    z[:, :, 0::2, 0::2] = x
    return _depthwise(z, filt.astype(x.dtype), (2, 1, 2, 1), stride=1)


def upsample2d_closed_form(x: np.ndarray) -> np.ndarray:
    """Polyphase form of a4 used by the HIP kernels: per axis
    out[2i] = g[i-1]/4 + 3 g[i]/4, out[2i+1] = 3 g[i]/4 + g[i+1]/4, zeros outside."""
    def axis(v, ax):
        v = np.moveaxis(v, ax, -1)
        p = np.pad(v, [(0, 0)] * (v.ndim - 1) + [(1, 1)])
        lo, mid, hi = p[..., :-2], p[..., 1:-1], p[..., 2:]
        out = np.empty(v.shape[:-1] + (2 * v.shape[-1],), dtype=v.dtype)
        q, t = v.dtype.type(0.25), v.dtype.type(0.75)
        out[..., 0::2] = q * lo + t * mid
        out[..., 1::2] = t * mid + q * hi
        return np.moveaxis(out, -1, ax)
    return axis(axis(x, 2), 3)


def pointwise(x: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray] = None) -> np.ndarray:
    """1x1 convolution (reference conv2 :130-136, fromrgb :186, torgb :268/:300)."""
    n, c, h, wd = x.shape
    co = w.shape[0]
    y = np.matmul(w.reshape(co, c).astype(x.dtype), x.reshape(n, c, h * wd)).reshape(n, co, h, wd)
    if bias is not None:
        y = y + bias.reshape(1, co, 1, 1).astype(x.dtype)
    return y


# --------------------------------------------------------------------------- a5
def separable_conv(x: np.ndarray, sd: Dict[str, np.ndarray], prefix: str, gemm16: bool = False) -> np.ndarray:
    """SeparableConv2d.forward (reference :154-170): dw3x3(+bias) -> act ->
    [FIR down] -> 1x1 -> [FIR up] -> [+ noise_const*noise_strength] -> act."""
    dt = x.dtype
    x = _depthwise(x, sd[f"{prefix}.conv1.weight"].astype(dt), (1, 1, 1, 1),
                   bias=sd[f"{prefix}.conv1.bias"])                     # :155
    x = lrelu_agc(x)                                                     # :156-157
    if f"{prefix}.downsample.filter.weight" in sd:
        x = downsample2d(x, sd[f"{prefix}.downsample.filter.weight"])    # :159-160
    w2 = sd[f"{prefix}.conv2.weight"]
    if gemm16:
        x, w2 = round_gemm_operands(x, w2)
    x = pointwise(x, w2)                                                 # :161
    if f"{prefix}.upsample.filter.weight" in sd:
        x = upsample2d(x, sd[f"{prefix}.upsample.filter.weight"])        # :162-163
    if f"{prefix}.noise_const" in sd:
        # fp32 product first, then the add (reference :166-167)
        noise = noise_plane(sd[f"{prefix}.noise_const"], x.shape[2], x.shape[3]).astype(dt) * sd[f"{prefix}.noise_strength"].astype(dt)
        x = x + noise
    return lrelu_agc(x)                                                  # :168-169


# --------------------------------------------------------------------------- a6/a7
def encoder(img: np.ndarray, sd: Dict[str, np.ndarray], resolution: int,
            taps: Optional[dict] = None, storage: Optional[str] = None, gemm16: bool = False):
    """Encoder.forward (reference :235-246) with EncoderBlock.forward (:192-200)."""
    feats = {}
    x = None
    res = resolution
    while res >= 4:
        b = f"encoder.b{res}"
        if f"{b}.fromrgb.weight" in sd:
            y = lrelu_agc(pointwise(img, sd[f"{b}.fromrgb.weight"], sd[f"{b}.fromrgb.bias"]))  # :194-195
            x = y if x is None else x + y                                 # :196
        feat = round_storage(separable_conv(x, sd, f"{b}.conv1", gemm16), storage)   # :198
        x = round_storage(separable_conv(feat, sd, f"{b}.conv2", gemm16), storage)   # :199
        feats[res] = feat
        if taps is not None:
            taps[f"{b}.conv1"] = feat
            taps[f"{b}.conv2"] = x
        res //= 2
    return x, feats


# --------------------------------------------------------------------------- a8/a9/a10
def synthesis(x: np.ndarray, feats: Dict[int, np.ndarray], sd: Dict[str, np.ndarray],
              resolution: int, taps: Optional[dict] = None, storage: Optional[str] = None, gemm16: bool = False) -> np.ndarray:
    """Synthesis.forward (reference :347-352), SynthesisBlockFirst (:270-279),
    SynthesisBlock (:303-315)."""
    img = None
    res = 4
    while res <= resolution:
        b = f"synthesis.b{res}"
        x = separable_conv(x, sd, f"{b}.conv1", gemm16)                   # :271 / :304
        if taps is not None:
            taps[f"{b}.conv1"] = x                                        # SeparableConv2d output (pre skip)
        x = round_storage(x + feats[res], storage)                        # :272 / :305
        if taps is not None:
            taps[f"{b}.conv1.skip"] = x
        x = round_storage(separable_conv(x, sd, f"{b}.conv2", gemm16), storage)   # :273 / :306
        if taps is not None:
            taps[f"{b}.conv2"] = x
        y = pointwise(x, sd[f"{b}.torgb.weight"], sd[f"{b}.torgb.bias"])  # :277 / :312
        if img is not None:
            img = upsample2d(img, sd[f"{b}.upsample.filter.weight"])      # :308-309
            img = img + y                                                 # :313
        else:
            img = y
        if taps is not None:
            taps[f"{b}.img"] = img
        res *= 2
    return img


def generator(x: np.ndarray, sd: Dict[str, np.ndarray], resolution: int,
              dtype=np.float32, taps: Optional[dict] = None, storage: Optional[str] = None, gemm16: Optional[bool] = None) -> np.ndarray:
    """Generator.forward (reference :362-369): x [N,4,R,R] -> img [N,3,R,R].
    x may be [N,4,H,W] with H, W multiples of R/4 (arbitrary-size forward, noise_plane above).
    storage: None / 'f32' = the reference; 'bf16' / 'f16' = the library's 16-bit activation storage modes.
    gemm16: the 1x1 convolutions on fp16-rounded operands (GEMM variant "f16", the default of the 16-bit storage modes;
    None = that default: on with 16-bit storage, off with fp32)."""
    x = np.asarray(x, dtype=dtype)
    sd = {k: np.asarray(v) for k, v in sd.items()}
    if gemm16 is None:
        gemm16 = storage in ("bf16", "f16")
    h, feats = encoder(x, sd, resolution, taps, storage, gemm16)
    return synthesis(h, feats, sd, resolution, taps, storage, gemm16)
