"""ORACLE (test infrastructure, never shipped, never on the product path).

Op-for-op restatement of ``lib/model_zoo/migan_inference.py`` on torch-CPU
(``F.conv2d`` / ``F.pad`` / nearest upsample, i.e. the oneDNN kernels the
reference's ``scripts/demo.py --device cpu`` path executes), driven directly by
a state_dict.  It exists for two jobs:

  * full-size parity (512x512, batch) in seconds instead of minutes of numpy;
  * the ``cpu_baseline`` leg of bench.py: the reference cannot travel to the GPU
    box (/root/reference is absent there), so this port is what gets timed on
    the host cores beside the GPU number (``cpu_baseline.kind == "port"``).

Pinned against the same golden vectors as oracle/migan_oracle.py
(tests/test_oracle_golden.py).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

_SQRT2 = float(np.sqrt(2))


def _act(x: torch.Tensor) -> torch.Tensor:
    # reference :20-28  (in-place leaky relu, fp32 multiply by sqrt(2), clamp +-256)
    x = F.leaky_relu(x, negative_slope=0.2, inplace=True)
    x = x * _SQRT2
    return x.clamp(-256.0, 256.0)


def _store(x: torch.Tensor, storage: Optional[str]) -> torch.Tensor:
    """16-bit activation storage modes of the HIP library (oracle/migan_oracle.py::round_storage): round to nearest even once
    per stored feature map, everything else fp32."""
    if storage in (None, "f32"):
        return x
    return x.to({"bf16": torch.bfloat16, "f16": torch.float16}[storage]).float()


def _zero_insertion_mask(sd, key: str, x: torch.Tensor) -> torch.Tensor:
    """filter_const (reference :85): 1 at even/even, 0 elsewhere; rebuilt at x's size for the arbitrary-size forward
    (reference README.md:87)."""
    fc = sd[key]
    if fc.shape[-2:] == x.shape[-2:]:
        return fc
    m = torch.zeros((1, 1) + tuple(x.shape[-2:]), dtype=x.dtype)
    m[:, :, 0::2, 0::2] = 1
    return m


def _noise(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    nc = sd[f"{p}.noise_const"]
    h, w = x.shape[-2:]
    if tuple(nc.shape) != (h, w):          # arbitrary-size forward: tiled periodically and cropped
        nc = nc.repeat((h + nc.shape[0] - 1) // nc.shape[0], (w + nc.shape[1] - 1) // nc.shape[1])[:h, :w]
    return nc * sd[f"{p}.noise_strength"]


def _gemm16(x: torch.Tensor, w: torch.Tensor):
    """GEMM variant "f16" (oracle/migan_oracle.py::round_gemm_operands): operands of the 1x1 rounded to fp16 after exact
    power-of-two scaling, fp32 products and sums."""
    xr = (x * 128.0).half().float() / 128.0
    m = float(w.abs().max())
    if m > 0 and np.isfinite(m):
        e = min(max(int(np.floor(np.log2(m))), -100), 100)
        s = float(2.0 ** (13 - e))
        w = (w * s).half().float() / s
    return xr, w


def _sepconv(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str, gemm16: bool = False) -> torch.Tensor:
    # reference :154-170
    c = x.shape[1]
    x = F.conv2d(x, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1, groups=c)
    x = _act(x)
    k = f"{p}.downsample.filter.weight"
    if k in sd:                                                  # :58-76
        x = F.conv2d(x, sd[k], None, stride=2, padding=1, groups=c)
    w2 = sd[f"{p}.conv2.weight"]
    if gemm16:
        x, w2 = _gemm16(x, w2)
    x = F.conv2d(x, w2)
    k = f"{p}.upsample.filter.weight"
    if k in sd:                                                  # :98-103
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = x * _zero_insertion_mask(sd, f"{p}.upsample.filter_const", x)
        x = F.pad(x, (2, 1, 2, 1))
        x = F.conv2d(x, sd[k], None, groups=x.shape[1])
    k = f"{p}.noise_const"
    if k in sd:                                                  # :165-167
        x = x.add_(_noise(sd, p, x))
    return _act(x)


@torch.no_grad()
def generator(x, sd, resolution: int, taps: Optional[dict] = None, storage: Optional[str] = None,
              gemm16: Optional[bool] = None) -> torch.Tensor:
    """x [N,4,R,R] float32 (tensor or ndarray; [N,4,H,W] with H, W multiples of R/4 for the arbitrary-size forward),
    sd: name -> tensor/ndarray.  storage: None / 'f32' = the reference; 'bf16' / 'f16' = 16-bit activation storage."""
    if gemm16 is None:          # GEMM variant "f16" is the default of the 16-bit storage modes
        gemm16 = storage in ("bf16", "f16")
    x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).float().cpu()
    sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))).float().cpu()
          for k, v in sd.items()}
    img_in = x
    feats = {}
    h = None
    res = resolution
    while res >= 4:                                              # :235-246
        b = f"encoder.b{res}"
        if f"{b}.fromrgb.weight" in sd:                          # :193-196
            y = _act(F.conv2d(img_in, sd[f"{b}.fromrgb.weight"], sd[f"{b}.fromrgb.bias"]))
            h = y if h is None else h + y
        feat = _store(_sepconv(h, sd, f"{b}.conv1", gemm16), storage)
        h = _store(_sepconv(feat, sd, f"{b}.conv2", gemm16), storage)
        feats[res] = feat
        if taps is not None:
            taps[f"{b}.conv1"] = feat
            taps[f"{b}.conv2"] = h
        res //= 2
    img = None
    res = 4
    while res <= resolution:                                     # :347-352
        b = f"synthesis.b{res}"
        h = _sepconv(h, sd, f"{b}.conv1", gemm16)
        if taps is not None:
            taps[f"{b}.conv1"] = h                               # SeparableConv2d output (pre skip)
        h = _store(h + feats[res], storage)                      # :272 / :305
        if taps is not None:
            taps[f"{b}.conv1.skip"] = h
        h = _store(_sepconv(h, sd, f"{b}.conv2", gemm16), storage)
        if taps is not None:
            taps[f"{b}.conv2"] = h
        y = F.conv2d(h, sd[f"{b}.torgb.weight"], sd[f"{b}.torgb.bias"])
        if img is not None:                                      # :308-313
            u = F.interpolate(img, scale_factor=2, mode="nearest")
            u = u * _zero_insertion_mask(sd, f"{b}.upsample.filter_const", u)
            u = F.pad(u, (2, 1, 2, 1))
            img = F.conv2d(u, sd[f"{b}.upsample.filter.weight"], None, groups=3).add_(y)
        else:
            img = y
        if taps is not None:
            taps[f"{b}.img"] = img
        res *= 2
    return img
