#!/usr/bin/env python3
"""Golden vectors that pin oracle/comodgan_oracle.py to the REFERENCE Co-Mod-GAN generator.

Run in the build container only (imports /root/reference, absent on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_comodgan.py

Outputs (committed):
    tests/golden/comodgan_schema.json    state_dict keys/shapes/kinds at 256 and 512 (reference constructors)
    tests/golden/comodgan_*.npz          whole-generator outputs (+ tap summaries of every block) of
                                         lib.model_zoo.comodgan.Generator with noise_mode='const', explicit z

Weights, inputs and z come from mi-gan_amd/synth.py (seeded, RNG-library independent), so the fixtures hold
outputs only.  The reference Synthesis defines num_ws only for 256/512 (comodgan.py:367-370); for the small
test geometries the attribute is set by hand before the Generator is assembled.
"""
import importlib
import json
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
cs = importlib.import_module("mi-gan_amd.comodgan_schema")
from lib.model_zoo.comodgan import Mapping, Encoder, Synthesis, Generator  # noqa: E402  (the reference itself)

torch.set_num_threads(max(1, os.cpu_count() or 1))

# (tag, resolution, ch_base, ch_max, batch, seed, truncation_psi[, truncation_cutoff])
CASES = [
    ("r16_c64", 16, 1024, 64, 2, 1, 1.0),
    ("r32_c128", 32, 4096, 128, 3, 2, 1.0),
    ("r32_c128_psi", 32, 4096, 128, 2, 3, 0.7),
    ("r64_c64", 64, 4096, 64, 2, 4, 1.0),
    ("r64_std", 64, 32768, 512, 1, 5, 1.0),       # the real channel rule (512 everywhere at <= 64)
    ("r256_small", 256, 16384, 256, 1, 6, 1.0),   # the 64..256-channel range at the big resolutions
    ("r512_std", 512, 32768, 512, 1, 7, 1.0),     # comodgan-512 as scripts/demo.py:101-106 builds it (BASELINE configs[4] geometry)
    ("r32_c128_cut3", 32, 4096, 128, 2, 8, 0.6, 3),   # truncation_cutoff: ws rows 0-2 truncated, 3-7 raw (round 3)
]


def tap_summary(t: torch.Tensor) -> np.ndarray:
    a = t.detach().double().numpy().ravel()
    idx = (np.arange(13, dtype=np.int64) * 2654435761 + 12345) % a.size
    return np.concatenate([[a.mean(), a.std(), np.abs(a).max()], a[idx]]).astype(np.float64)


def build(cfg):
    m = Mapping(num_ws=cfg.num_ws)
    e = Encoder(resolution=cfg.resolution, ch_base=cfg.ch_base, ch_max=cfg.ch_max)
    s = Synthesis(resolution=cfg.resolution, ch_base=cfg.ch_base, ch_max=cfg.ch_max)
    s.num_ws = cfg.num_ws
    return Generator(m, e, s).eval()


def schema_json():
    out = {}
    for r in (256, 512):
        cfg = cs.Config(resolution=r, num_ws=cs.default_num_ws(r))
        g = build(cfg)
        params = {k for k, _ in g.named_parameters()}
        out[str(r)] = sorted([k, list(v.shape), "param" if k in params else "buffer"] for k, v in g.state_dict().items())
    with open(os.path.join(HERE, "comodgan_schema.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("comodgan_schema.json", {k: len(v) for k, v in out.items()})


def main():
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None      # regenerate one case, leave the others untouched
    if only is None:
        schema_json()
    for case in CASES:
        tag, r, cb, cm, n, seed, psi = case[:7]
        cutoff = case[7] if len(case) > 7 else None
        if only is not None and tag != only:
            continue
        cfg = cs.Config(resolution=r, ch_base=cb, ch_max=cm, num_ws=cs.default_num_ws(r))
        sd = synth.make_comodgan_state_dict(cfg, seed)
        g = build(cfg)
        g.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        x = synth.make_input(n, r, seed)
        z = synth.make_latent(n, cfg.z_dim, seed)
        taps = {}
        hooks = []
        for name, mod in g.named_modules():
            if name.startswith(("encoder.b", "synthesis.b")) and name.count(".") == 1:
                def hook(_m, _i, o, name=name):
                    for j, t in enumerate(o if isinstance(o, tuple) else (o,)):
                        if torch.is_tensor(t):
                            taps[f"{name}/{j}"] = tap_summary(t)
                hooks.append(mod.register_forward_hook(hook))
        with torch.no_grad():
            y = g(torch.from_numpy(x), z=torch.from_numpy(z), truncation_psi=psi, truncation_cutoff=cutoff, noise_mode="const")
        for h in hooks:
            h.remove()
        out = {"y": y.numpy().astype(np.float32), "cfg": np.asarray([r, cb, cm, n, seed], dtype=np.int64),
               "psi": np.asarray(psi), "cutoff": np.asarray(-1 if cutoff is None else cutoff)}
        out.update({"tap:" + k: v for k, v in taps.items()})
        np.savez_compressed(os.path.join(HERE, f"comodgan_{tag}.npz"), **out)
        print(tag, "y", y.shape, "absmax %.3f" % y.abs().max().item(), "taps", len(taps))


if __name__ == "__main__":
    main()
