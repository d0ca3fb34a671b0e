"""The Co-Mod-GAN oracle (oracle/comodgan_oracle.py) and the schema table are held to the outputs of the
reference module itself (tests/golden/comodgan_*.npz, comodgan_schema.json; made by make_golden_comodgan.py)."""
import glob
import importlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import comodgan_oracle as orc

pkg = importlib.import_module("mi-gan_amd")
cs = importlib.import_module("mi-gan_amd.comodgan_schema")
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(os.path.basename(p)[len("comodgan_"):-4] for p in glob.glob(os.path.join(GOLD, "comodgan_*.npz")))


def load_case(tag):
    g = np.load(os.path.join(GOLD, f"comodgan_{tag}.npz"))
    r, cb, cm, n, seed = (int(v) for v in g["cfg"])
    cfg = cs.Config(resolution=r, ch_base=cb, ch_max=cm, num_ws=cs.default_num_ws(r))
    sd = pkg.synth.make_comodgan_state_dict(cfg, seed)
    x = pkg.synth.make_input(n, r, seed)
    z = pkg.synth.make_latent(n, cfg.z_dim, seed)
    cutoff = int(g["cutoff"]) if "cutoff" in g.files and int(g["cutoff"]) >= 0 else None      # (goldens before round 3 have no cutoff)
    return g, cfg, sd, x, z, (float(g["psi"]), cutoff)


def tap_summary(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    idx = (np.arange(13, dtype=np.int64) * 2654435761 + 12345) % a.size
    return np.concatenate([[a.mean(), a.std(), np.abs(a).max()], a[idx]])


def test_schema_matches_reference_constructors():
    with open(os.path.join(GOLD, "comodgan_schema.json")) as f:
        ref = json.load(f)
    for r in (256, 512):
        cfg = cs.Config(resolution=r, num_ws=cs.default_num_ws(r))
        mine = sorted([e.name, list(e.shape), e.kind] for e in cs.entries(cfg))
        assert mine == ref[str(r)]
    assert cs.default_num_ws(256) == 14 and cs.default_num_ws(512) == 16     # comodgan.py:367-370


@pytest.mark.parametrize("tag", CASES)
def test_oracle_reproduces_reference_outputs(tag):
    g, cfg, sd, x, z, (psi, cutoff) = load_case(tag)
    taps = {}
    y = orc.generator(x, z, sd, cfg.resolution, cfg.num_ws, truncation_psi=psi, truncation_cutoff=cutoff, taps=taps)
    ref = g["y"]
    scale = np.abs(ref).max()
    assert np.abs(y - ref).max() <= 3e-5 * scale, (np.abs(y - ref).max(), scale)
    # block outputs (x, feat / x, img): reference forward hooks vs the oracle's taps
    checked = 0
    for key in g.files:
        if not key.startswith("tap:"):
            continue
        name, j = key[4:].split("/")
        want = g[key]
        if name.startswith("encoder.b") and name != "encoder.b4":
            got = taps[name + (".conv1" if j == "0" else ".conv0")]      # (x, img=None, feat): tuple index 0 and 2
        elif name == "encoder.b4":
            got = taps["encoder.b4.fc" if j == "0" else "encoder.b4.conv"]
        elif name == "synthesis.b4":
            got = taps["synthesis.b4.conv" if j == "0" else "synthesis.b4.img"]
        else:
            if j == "2":
                continue                                                # to_rgb_out (img increment), not tapped
            got = taps[name + (".conv1" if j == "0" else ".img")]
        s = tap_summary(got)
        assert np.allclose(s, want, rtol=0, atol=4e-5 * max(1.0, want[2])), (key, s[:3], want[:3])
        checked += 1
    assert checked >= 10


def test_float64_oracle_bounds_fp32_rounding():
    g, cfg, sd, x, z, psi = load_case("r32_c128")
    y32 = orc.generator(x, z, sd, cfg.resolution, cfg.num_ws)
    y64 = orc.generator(x, z, sd, cfg.resolution, cfg.num_ws, dtype=torch.float64)
    assert np.abs(y32 - y64).max() <= 1e-4 * np.abs(y64).max()
