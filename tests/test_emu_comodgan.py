"""CPU execution of the Co-Mod-GAN product kernels + host plan + C ABI (include/comodgan_hip.h) through the fiber SIMT
emulator (tests/emu), compared with the oracle and the reference's golden outputs.  No GPU involved."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

from oracle import comodgan_oracle as orc
from tests.emu_util import aligned, emu_lib

pkg = importlib.import_module("mi-gan_amd")
cs = importlib.import_module("mi-gan_amd.comodgan_schema")
hb = pkg.hipbind
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_emu(cfg, sd, x, z, psi=1.0, noise_mode="const", noise=None, debug=()):
    lib = emu_lib()
    h = hb.CoModGANHandle(lib, cfg.resolution, cfg.num_ws, cfg.ch_base, cfg.ch_max, cfg.z_dim, cfg.w_dim, cfg.w0_dim, cfg.map_layers)
    keep = {k: aligned(v) for k, v in sd.items()}
    for name, shape, _ in h.weights():
        assert tuple(sd[name].shape) == tuple(shape), name
        h.set_weight(name, keep[name].ctypes.data, shape)
    h.commit()
    if debug:
        h.set_debug(True)
    n = x.shape[0]
    nbytes = h.workspace_bytes(n)
    ws = np.zeros(nbytes // 4 + 64, dtype=np.float32)
    off = (256 - ws.ctypes.data % 256) % 256 // 4
    wsv = ws[off:]
    xa, za = aligned(x), aligned(z)
    y = aligned(np.zeros((n, 3, cfg.resolution, cfg.resolution), np.float32))
    na = aligned(noise) if noise is not None else None
    h.forward(xa.ctypes.data, za.ctypes.data, y.ctypes.data, n, wsv.ctypes.data, nbytes, truncation_psi=psi, noise_mode=noise_mode,
              noise_ptr=None if na is None else na.ctypes.data)
    taps = {}
    for name in debug:
        o, shp = h.debug_tensor(n, name)
        taps[name] = wsv[o // 4:o // 4 + int(np.prod(shp))].reshape(shp).copy()
    info = h.launches()
    h.close()
    return y, taps, info


def case(tag):
    g = np.load(os.path.join(GOLD, f"comodgan_{tag}.npz"))
    r, cb, cm, n, seed = (int(v) for v in g["cfg"])
    cfg = cs.Config(resolution=r, ch_base=cb, ch_max=cm, num_ws=cs.default_num_ws(r))
    return g, cfg, pkg.synth.make_comodgan_state_dict(cfg, seed), pkg.synth.make_input(n, r, seed), pkg.synth.make_latent(n, cfg.z_dim, seed)


def test_schema_of_the_c_abi_matches_the_python_table():
    lib = emu_lib()
    for r, cb, cm in ((512, 32768, 512), (256, 32768, 512), (16, 1024, 64)):
        cfg = cs.Config(resolution=r, ch_base=cb, ch_max=cm, num_ws=cs.default_num_ws(r))
        h = hb.CoModGANHandle(lib, r, cfg.num_ws, cb, cm)
        mine = sorted((n, tuple(s), b) for n, s, b in h.weights())
        want = sorted((e.name, tuple(e.shape), e.kind == "buffer") for e in cs.entries(cfg))
        assert mine == want
        h.close()


def test_generator_r16_layers_and_output():
    g, cfg, sd, x, z = case("r16_c64")
    names = ["mapping", "encoder.b16.conv0", "encoder.b16.conv1", "encoder.b8.conv0", "encoder.b8.conv1", "encoder.b4.conv",
             "encoder.b4.fc", "synthesis.b4.conv", "synthesis.b4.img", "synthesis.b8.conv0", "synthesis.b8.conv1", "synthesis.b8.img",
             "synthesis.b16.conv0", "synthesis.b16.conv1"]
    y, taps, info = run_emu(cfg, sd, x, z, debug=names)
    want_taps = {}
    want = orc.generator(x, z, sd, cfg.resolution, cfg.num_ws, taps=want_taps)
    for name in names:
        w = want_taps[name]
        got = taps[name]
        if w.ndim == 4 and not name.endswith(".img"):
            got = np.transpose(got, (0, 3, 1, 2))
        err = np.abs(got - w).max()
        assert err <= 2e-4 * max(1.0, np.abs(w).max()), (name, err, np.abs(w).max())
    assert np.abs(y - want).max() <= 1e-3
    assert np.abs(y - g["y"]).max() <= 1e-3          # the reference module's own output
