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


def _sepconv(x: torch.Tensor, sd: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    # reference :154-170
    c = x.shape[1]
    x = F.conv2d(x, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1, groups=c)
    x = _act(x)
    k = f"{p}.downsample.filter.weight"
    if k in sd:                                                  # :58-76
        x = F.conv2d(x, sd[k], None, stride=2, padding=1, groups=c)
    x = F.conv2d(x, sd[f"{p}.conv2.weight"])
    k = f"{p}.upsample.filter.weight"
    if k in sd:                                                  # :98-103
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = x * sd[f"{p}.upsample.filter_const"]
        x = F.pad(x, (2, 1, 2, 1))
        x = F.conv2d(x, sd[k], None, groups=x.shape[1])
    k = f"{p}.noise_const"
    if k in sd:                                                  # :165-167
        x = x.add_(sd[k] * sd[f"{p}.noise_strength"])
    return _act(x)


@torch.no_grad()
def generator(x, sd, resolution: int, taps: Optional[dict] = None) -> torch.Tensor:
    """x [N,4,R,R] float32 (tensor or ndarray), sd: name -> tensor/ndarray."""
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
        feat = _sepconv(h, sd, f"{b}.conv1")
        h = _sepconv(feat, sd, f"{b}.conv2")
        feats[res] = feat
        if taps is not None:
            taps[f"{b}.conv1"] = feat
            taps[f"{b}.conv2"] = h
        res //= 2
    img = None
    res = 4
    while res <= resolution:                                     # :347-352
        b = f"synthesis.b{res}"
        h = _sepconv(h, sd, f"{b}.conv1")
        if taps is not None:
            taps[f"{b}.conv1"] = h                               # SeparableConv2d output (pre skip)
        h = h + feats[res]                                       # :272 / :305
        if taps is not None:
            taps[f"{b}.conv1.skip"] = h
        h = _sepconv(h, sd, f"{b}.conv2")
        if taps is not None:
            taps[f"{b}.conv2"] = h
        y = F.conv2d(h, sd[f"{b}.torgb.weight"], sd[f"{b}.torgb.bias"])
        if img is not None:                                      # :308-313
            u = F.interpolate(img, scale_factor=2, mode="nearest")
            u = u * sd[f"{b}.upsample.filter_const"]
            u = F.pad(u, (2, 1, 2, 1))
            img = F.conv2d(u, sd[f"{b}.upsample.filter.weight"], None, groups=3).add_(y)
        else:
            img = y
        if taps is not None:
            taps[f"{b}.img"] = img
        res *= 2
    return img
