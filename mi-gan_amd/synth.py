"""Deterministic synthetic weights and inputs for tests, smoke() and bench.py.

There is no network and the reference ships no checkpoints (reference
.gitignore:19, README.md:50-51), so every number in this repository is produced
from seeded synthetic data.  The generator below is a counter-based hash
(splitmix64) + Box-Muller written in numpy so the streams do not depend on the
torch / numpy RNG implementation.

Weight regimes:
  'export'  rows L2-normalised like scripts/export_inference_model.py:18-27,
            biases ~N(0,0.5^2), noise_strength ~N(0,0.3^2), noise_const ~N(0,1)
            -> output |max| of order 10, noise and clamp paths exercised.
  'init'    small uniform weights, noise_strength = 0 (the constructor's state).
Inputs follow scripts/demo.py:56-66: x = cat([mask-0.5, img*mask]).
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

from . import schema

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(idx: np.ndarray, key: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(key & 0xFFFFFFFFFFFFFFFF)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _key(seed: int, tag: str) -> int:
    return (int(seed) * 0x100000001B3 + zlib.crc32(tag.encode())) & 0xFFFFFFFFFFFFFFFF


def uniform(shape, seed: int, tag: str) -> np.ndarray:
    """U[0,1) float64, reproducible from (seed, tag)."""
    n = int(np.prod(shape)) if len(shape) else 1
    z = _splitmix64(np.arange(n, dtype=np.uint64), _key(seed, tag))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def normal(shape, seed: int, tag: str) -> np.ndarray:
    """N(0,1) float64 via Box-Muller on two hashed uniform streams."""
    u1 = uniform(shape, seed, tag + "/a")
    u2 = uniform(shape, seed, tag + "/b")
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def make_state_dict(resolution: int, seed: int = 0, regime: str = "export") -> Dict[str, np.ndarray]:
    """Reference-shaped state_dict (numpy float32) for ``Generator(resolution)``."""
    out: Dict[str, np.ndarray] = {}
    for e in schema.entries(resolution):
        shp = e.shape
        if e.role in ("dw_w", "pw_w", "rgb_w"):
            if regime == "export":
                w = normal(shp, seed, e.name)
                # per-output-channel L2 normalisation over (Ci,kh,kw):
                # scripts/export_inference_model.py:26
                nrm = np.sqrt((w * w).reshape(shp[0], -1).sum(1) + 1e-8).reshape(-1, 1, 1, 1)
                w = w / nrm
            else:
                fan_in = int(np.prod(shp[1:]))
                w = (uniform(shp, seed, e.name) * 2.0 - 1.0) / np.sqrt(fan_in)
            out[e.name] = w.astype(np.float32)
        elif e.role in ("dw_b", "rgb_b"):
            scale = 0.5 if regime == "export" else 0.05
            out[e.name] = (normal(shp, seed, e.name) * scale).astype(np.float32)
        elif e.role == "noise_strength":
            v = normal((1,), seed, e.name)[0] * 0.3 if regime == "export" else 0.0
            out[e.name] = np.asarray(v, dtype=np.float32).reshape(())
        elif e.role == "noise_const":
            out[e.name] = normal(shp, seed, e.name).astype(np.float32)
        elif e.role == "fir_down":
            k = np.asarray(schema.fir_kernel_2d(1.0), dtype=np.float32)
            out[e.name] = np.broadcast_to(k, shp).copy()
        elif e.role == "fir_up":
            k = np.asarray(schema.fir_kernel_2d(4.0), dtype=np.float32)
            out[e.name] = np.broadcast_to(k, shp).copy()
        elif e.role == "filter_const":
            fc = np.zeros(shp, dtype=np.float32)
            fc[..., 0::2, 0::2] = 1.0          # reference :83-85
            out[e.name] = fc
        else:  # pragma: no cover
            raise AssertionError(e.role)
    return out


def make_masks(batch: int, resolution: int, seed: int = 0) -> np.ndarray:
    """Free-form-like {0,1} masks [N,1,R,R] (1 = known pixel, demo.py:60):
    a few random rectangles and thick strokes punched out of an all-ones mask,
    in the spirit of scripts/generate_masks.py:72-93 without the PIL dependency."""
    r = resolution
    m = np.ones((batch, 1, r, r), dtype=np.float32)
    u = uniform((batch, 8, 5), seed, f"mask{r}")
    for b in range(batch):
        for j in range(8):
            kind, a, c, d, e = u[b, j]
            if j >= 3 and kind < 0.35:
                continue
            h = max(1, int(a * r * (0.5 if j < 4 else 0.12)))
            w = max(1, int(c * r * (0.12 if j < 4 else 0.5)))
            if j % 2:
                h, w = w, h
            y0 = int(d * (r - 1)) - h // 2
            x0 = int(e * (r - 1)) - w // 2
            m[b, 0, max(y0, 0):min(y0 + h, r), max(x0, 0):min(x0 + w, r)] = 0.0
    return m


def make_input(batch: int, resolution: int, seed: int = 0, kind: str = "demo") -> np.ndarray:
    """Network input [N,4,R,R] float32.

    kind='demo'  : img_u8 ~ U{0..255} -> img*2/255-1, x = cat([mask-0.5, img*mask])
                   (scripts/demo.py:56-66)
    kind='randn' : N(0,1) in all four planes (calculate_flops.py:115 style stress input)
    """
    r = resolution
    if kind == "randn":
        return normal((batch, 4, r, r), seed, f"xin{r}").astype(np.float32)
    img_u8 = np.floor(uniform((batch, 3, r, r), seed, f"img{r}") * 256.0)
    img = (img_u8.astype(np.float32) * 2.0 / 255.0 - 1.0).astype(np.float32)
    mask = make_masks(batch, r, seed)
    return np.concatenate([mask - 0.5, img * mask], axis=1).astype(np.float32)


def make_uint8_input(batch: int, resolution: int, seed: int = 0):
    """The uint8 source of make_input(kind='demo'): image [N,R,R,3] uint8 (HWC, as np.array of a PIL image, demo.py:59) and
    mask [N,R,R] uint8 (255 = known pixel, demo.py:60); preprocess(image, mask) == make_input(batch, resolution, seed)."""
    r = resolution
    img_u8 = np.floor(uniform((batch, 3, r, r), seed, f"img{r}") * 256.0).astype(np.uint8)
    mask = make_masks(batch, r, seed)
    return np.ascontiguousarray(np.transpose(img_u8, (0, 2, 3, 1))), (mask[:, 0] * 255.0).astype(np.uint8)


# ---- Co-Mod-GAN -------------------------------------------------------

def make_comodgan_state_dict(cfg, seed: int = 0) -> Dict[str, np.ndarray]:
    """Reference-shaped state_dict (numpy float32) for the Co-Mod-GAN generator of ``cfg``
    (comodgan_schema.Config).  Weights ~N(0,1) like the constructors (stylegan.py:79,213; the mapping's
    are divided by lr_multiplier=0.01, :79,:366), biases, noise strengths and w_avg non-zero so every
    term of the forward is exercised; affine biases around 1 (stylegan.py:273)."""
    from . import comodgan_schema as cs
    out: Dict[str, np.ndarray] = {}
    for e in cs.entries(cfg):
        shp, tag = e.shape, "cm/" + e.name
        mapping = e.name.startswith("mapping.")
        if e.role in ("conv_w", "rgb_w", "affine_w"):
            out[e.name] = normal(shp, seed, tag).astype(np.float32)
        elif e.role == "dense_w":
            out[e.name] = (normal(shp, seed, tag) * (100.0 if mapping else 1.0)).astype(np.float32)
        elif e.role in ("conv_b", "rgb_b"):
            out[e.name] = (normal(shp, seed, tag) * 0.3).astype(np.float32)
        elif e.role == "dense_b":
            out[e.name] = (normal(shp, seed, tag) * (30.0 if mapping else 0.3)).astype(np.float32)
        elif e.role == "affine_b":
            out[e.name] = (1.0 + normal(shp, seed, tag) * 0.3).astype(np.float32)
        elif e.role == "noise_strength":
            out[e.name] = np.asarray(normal((1,), seed, tag)[0] * 0.3, dtype=np.float32).reshape(())
        elif e.role == "noise_const":
            out[e.name] = normal(shp, seed, tag).astype(np.float32)
        elif e.role == "w_avg":
            out[e.name] = (normal(shp, seed, tag) * 0.3).astype(np.float32)
        elif e.role == "fir":
            out[e.name] = np.asarray(cs.fir_kernel_2d(), dtype=np.float32)
        else:  # pragma: no cover
            raise AssertionError(e.role)
    return out


def make_latent(batch: int, z_dim: int = 512, seed: int = 0) -> np.ndarray:
    """z [N, z_dim] ~ N(0,1) (comodgan.py:438-439 draws it with torch.randn; parity needs it explicit)."""
    return normal((batch, z_dim), seed, f"z{z_dim}").astype(np.float32)
