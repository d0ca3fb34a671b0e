#!/usr/bin/env python
"""Golden pair for the checkpoint converter: a random training-style weight tree and the
inference state_dict the REFERENCE's own copy_weights() (scripts/export_inference_model.py:17-85) makes of it.

copy_weights only reads attributes (.reparametrize, .num_reparam_tensors, .w0.., .weight, .bias, .noise_const,
.noise_strength), so the source is a duck-typed tree of random tensors shaped after the reference inference
Generator; the destination is the reference's migan_inference.Generator itself.  Build container only.

    python tests/golden/make_golden_convert.py    -> tests/golden/convert_r16.npz
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
for name in ("cv2", "torchvision", "torchvision.transforms"):         # imported at module level, unused by copy_weights
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
if not hasattr(sys.modules["torchvision"], "transforms"):
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
spec = importlib.util.spec_from_file_location("ref_export", os.path.join(REF, "scripts", "export_inference_model.py"))
ref_export = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_export)
from lib.model_zoo.migan_inference import Generator as RefGenerator     # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
pkg = importlib.import_module("mi-gan_amd")
SEED = 7


def rnd(shape, tag, scale=1.0):
    """training tensors come from mi-gan_amd/synth.py (library independent), so the fixture holds outputs only"""
    return torch.from_numpy((pkg.synth.normal(tuple(shape), SEED, tag) * scale).astype(np.float32))

R = 16
torch.manual_seed(7)
dest = RefGenerator(resolution=R)
flat = {}          # training-style state_dict


class Node(types.SimpleNamespace):
    pass


def conv(prefix, shape, bias, reparam, n=4):
    m = Node(reparametrize=reparam, num_reparam_tensors=n)
    if reparam:
        for i in range(n):
            w = rnd(shape, f"{prefix}.w{i}")
            setattr(m, f"w{i}", torch.nn.Parameter(w))
            flat[f"{prefix}.w{i}"] = w
    else:
        m.weight = torch.nn.Parameter(rnd(shape, f"{prefix}.weight"))
        flat[f"{prefix}.weight"] = m.weight.detach()
    m.bias = None
    if bias:
        m.bias = torch.nn.Parameter(rnd((shape[0],), f"{prefix}.bias", 0.1))
        flat[f"{prefix}.bias"] = m.bias.detach()
    return m


def sep(prefix, dmod, noise_res=None, flip=False):
    m = Node()
    m.conv1 = conv(f"{prefix}.conv1", tuple(dmod.conv1.weight.shape), dmod.conv1.bias is not None, reparam=not flip)
    m.conv2 = conv(f"{prefix}.conv2", tuple(dmod.conv2.weight.shape), dmod.conv2.bias is not None, reparam=True, n=3 if flip else 4)
    if noise_res is not None:
        m.conv2.noise_const = rnd((noise_res, noise_res), f"{prefix}.conv2.noise_const")
        m.conv2.noise_strength = torch.nn.Parameter(rnd((1,), f"{prefix}.conv2.noise_strength", 0.3).reshape(()))
        flat[f"{prefix}.conv2.noise_const"] = m.conv2.noise_const
        flat[f"{prefix}.conv2.noise_strength"] = m.conv2.noise_strength.detach()
    return m


src = Node(encoder=Node(), synthesis=Node())
for res in [2 ** i for i in range(2, int(np.log2(R)) + 1)]:
    d = getattr(dest.encoder, f"b{res}")
    b = Node()
    if d.fromrgb is not None:
        b.fromrgb = conv(f"encoder.b{res}.fromrgb", tuple(d.fromrgb.weight.shape), d.fromrgb.bias is not None, reparam=True)
    b.conv1 = sep(f"encoder.b{res}.conv1", d.conv1)
    b.conv2 = sep(f"encoder.b{res}.conv2", d.conv2, flip=(res == 8))
    setattr(src.encoder, f"b{res}", b)
    d = getattr(dest.synthesis, f"b{res}")
    b = Node()
    b.torgb = conv(f"synthesis.b{res}.torgb", tuple(d.torgb.weight.shape), d.torgb.bias is not None, reparam=(res != 4))
    b.conv1 = sep(f"synthesis.b{res}.conv1", d.conv1, noise_res=res if d.conv1.use_noise else None)
    b.conv2 = sep(f"synthesis.b{res}.conv2", d.conv2, noise_res=res if d.conv2.use_noise else None)
    setattr(src.synthesis, f"b{res}", b)

with torch.no_grad():
    ref_export.copy_weights(src, dest, resolution=R)                     # the reference function itself
out = {k: v.detach().cpu().numpy() for k, v in dest.state_dict().items()}
# recipe of the training tree (which convs are re-parameterised, with how many tensors) + the expected tensors:
# whole when small, else every 97th element plus float64 sum and sum of squares
recipe = sorted(flat.keys())
fix = {"recipe": np.array(recipe), "seed": SEED, "resolution": R}
for k, v in out.items():
    f = v.reshape(-1)
    if f.size <= 4096:
        fix["full/" + k] = v
    else:
        fix["samp/" + k] = f[::97].copy()
        fix["stat/" + k] = np.array([f.astype(np.float64).sum(), (f.astype(np.float64) ** 2).sum()])
np.savez_compressed(os.path.join(HERE, f"convert_r{R}.npz"), **fix)
print("wrote", f"convert_r{R}.npz", len(flat), "training tensors ->", len(out), "inference tensors")
