#!/usr/bin/env python3
"""Golden vectors for the arbitrary-size forward, from the REFERENCE module.

The reference is fixed-size only because of two registered buffers (README.md:87: "make `filter_const` and
`noise_const` computations dynamic"): this script runs the reference Generator(R) on an H x W input after replacing,
in every SeparableConv2d / Upsample2d instance, `filter_const` by the same even/even zero-insertion mask at the size
that instance sees (reference :85) and `noise_const` by itself tiled periodically and cropped to that size (:149).
Nothing else of the module is touched.  Run in the build container only (imports /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hw.py

Outputs (committed): tests/golden/generator_hw_*.npz (network outputs; weights and inputs come from mi-gan_amd/synth.py).
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MIGAN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

pkg = importlib.import_module("mi-gan_amd")
synth = pkg.synth
import lib.model_zoo.migan_inference as ref  # noqa: E402  (the reference module itself)

torch.set_num_threads(max(1, os.cpu_count() or 1))


def make_dynamic(g, resolution, height, width):
    """replace the fixed-size buffers of the reference module by ones of the size each layer sees at height x width"""
    for name, mod in g.named_modules():
        parts = name.split(".")
        blk = next((p for p in parts if p.startswith("b") and p[1:].isdigit()), None)
        if blk is None:
            continue
        res = int(blk[1:])                       # block b<res>: buffers are registered at res x res (reference :85, :149)
        h, w = height * res // resolution, width * res // resolution
        if isinstance(mod, ref.Upsample2d):
            m = torch.zeros(1, 1, h, w)
            m[:, :, 0::2, 0::2] = 1              # w.repeat(...) of [[1,0],[0,0]] (reference :83-85)
            assert torch.equal(m[:, :, :min(h, res), :min(w, res)], mod.filter_const[:, :, :min(h, res), :min(w, res)])
            mod.filter_const = m
        if isinstance(mod, ref.SeparableConv2d) and mod.use_noise:
            nc = mod.noise_const
            mod.noise_const = nc.repeat((h + res - 1) // res, (w + res - 1) // res)[:h, :w].contiguous()


def case(tag, resolution, height, width, batch, seed):
    sd_np = synth.make_state_dict(resolution, seed=seed, regime="export")
    x = (synth.normal((batch, 4, height, width), seed, "xhw") * 0.7).astype(np.float32)
    g = ref.Generator(resolution=resolution)
    g.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    g.eval()
    make_dynamic(g, resolution, height, width)
    with torch.no_grad():
        y = g(torch.from_numpy(x.copy())).numpy()
    assert y.shape == (batch, 3, height, width)
    np.savez_compressed(os.path.join(HERE, f"generator_hw_{tag}.npz"), resolution=resolution, height=height, width=width,
                        batch=batch, seed=seed, y=y.astype(np.float32), y_absmax=float(np.abs(y).max()))
    print(tag, y.shape, "absmax", float(np.abs(y).max()))


if __name__ == "__main__":
    case("r16_12x20", 16, 12, 20, 2, 11)        # smaller than R in one axis, larger in the other (noise tiled and cropped)
    case("r16_4x8", 16, 4, 8, 2, 11)            # the smallest size: the b4 level is 1 x 2
    case("r32_24x40", 32, 24, 40, 2, 11)
    case("r64_48x80", 64, 48, 80, 2, 91)
    case("r256_192x320", 256, 192, 320, 1, 91)
