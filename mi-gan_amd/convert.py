"""Checkpoint re-parameterisation: training-time MI-GAN generator weights -> the inference ``state_dict``
that ``Generator.load_state_dict`` (and the reference's ``migan_inference.Generator``) takes.

Mirror of ``copy_weights`` in the reference's scripts/export_inference_model.py:17-85 on plain state_dicts
(the reference walks live module trees of a pickled training snapshot):

  * re-parameterised convolutions (``w0 .. w{n-1}``): w = (w0 + w1 + ...) / sqrt(n)            (:19-23)
  * every convolution weight is L2-normalised per output channel: w * rsqrt(sum(w^2) + 1e-8)     (:26)
  * biases are copied as float32                                                                  (:37-50)
  * noise_const / noise_strength of a synthesis SeparableConv2d come from its second (1x1) conv   (:69-71, :81-83)
  * the FIR buffers of the inference model are constants and keep their constructor values.

Host-side tool (runs once per checkpoint, on CPU tensors); the forward itself stays GPU-only.
"""
from __future__ import annotations

import math
import re
from collections import OrderedDict
from typing import Mapping

import torch

from . import schema

_CONV_SUFFIX = re.compile(r"^(.*)\.(?:weight|w\d+)$")


def _source_weight(sd: Mapping[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """get_source_w of the reference (:18-27) for the training conv at ``prefix``."""
    if f"{prefix}.w0" in sd:
        n = 0
        while f"{prefix}.w{n}" in sd:
            n += 1
        w = sd[f"{prefix}.w0"]
        for i in range(1, n):
            w = w + sd[f"{prefix}.w{i}"]
        w = w / math.sqrt(n)
    elif f"{prefix}.weight" in sd:
        w = sd[f"{prefix}.weight"]
    else:
        raise KeyError(f"training checkpoint has neither {prefix}.weight nor {prefix}.w0")
    return w * (w.square().sum(dim=[1, 2, 3]) + 1e-8).rsqrt().reshape(-1, 1, 1, 1)


def convert_training_state_dict(train_sd: Mapping[str, torch.Tensor], resolution: int) -> "OrderedDict[str, torch.Tensor]":
    """Training generator ``state_dict`` (keys like ``encoder.b512.conv1.conv1.w0``) -> inference ``state_dict``
    with exactly the keys / shapes of ``Generator(resolution).state_dict()``."""
    schema.check_resolution(resolution)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for e in schema.entries(resolution):
        name, role = e.name, e.role
        if role in ("dw_w", "pw_w", "rgb_w"):
            t = _source_weight(train_sd, name[: -len(".weight")])
        elif role in ("dw_b", "rgb_b"):
            t = train_sd[name].to(torch.float32)
        elif role in ("noise_const", "noise_strength"):
            # inference SeparableConv2d.noise_* <- training SeparableConv2d.conv2.noise_*
            head, leaf = name.rsplit(".", 1)
            t = train_sd[f"{head}.conv2.{leaf}"]
        elif role == "fir_down":
            t = torch.tensor(schema.fir_kernel_2d(1.0)).repeat(e.shape[0], 1, 1, 1)
        elif role == "fir_up":
            t = torch.tensor(schema.fir_kernel_2d(4.0)).repeat(e.shape[0], 1, 1, 1)
        elif role == "filter_const":
            t = torch.tensor([[1.0, 0.0], [0.0, 0.0]]).repeat(1, 1, e.shape[2] // 2, e.shape[3] // 2)
        else:  # pragma: no cover
            raise AssertionError(role)
        t = t.detach().to(torch.float32).contiguous()
        if tuple(t.shape) != tuple(e.shape):
            raise ValueError(f"{name}: converted shape {tuple(t.shape)} does not match the inference model {tuple(e.shape)}")
        out[name] = t
    return out
