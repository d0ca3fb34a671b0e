"""Co-Mod-GAN parity on a real MI355X: comodgan.Generator (nn.Module -> C ABI -> HIP kernels) against the committed
outputs of the reference module and against the CPU oracle.  Tolerance: the north star's 1e-3 max-abs in fp32."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import comodgan_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("gpu tests need an MI355X (torch.cuda.is_available() is False)")
    return torch.device("cuda", 0)


def _build(pkg, cfg, seed, dev):
    cm = pkg.comodgan
    kw = dict(ch_base=cfg.ch_base, ch_max=cfg.ch_max)
    m = cm.Generator(cm.Mapping(num_ws=cfg.num_ws), cm.Encoder(resolution=cfg.resolution, **kw), cm.Synthesis(resolution=cfg.resolution, **kw))
    sd = pkg.synth.make_comodgan_state_dict(cfg, seed)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


def _cfg(pkg, r, cb=32768, cm=512):
    cs = pkg.comodgan_schema
    return cs.Config(resolution=r, ch_base=cb, ch_max=cm, num_ws=cs.default_num_ws(r))


CASES = sorted(os.path.basename(p)[len("comodgan_"):-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "comodgan_*.npz")))


@pytest.mark.parametrize("tag", CASES)
def test_reference_golden_outputs(pkg, dev, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"comodgan_{tag}.npz"))
    r, cb, cmx, n, seed = (int(v) for v in g["cfg"])
    cfg = _cfg(pkg, r, cb, cmx)
    m, _ = _build(pkg, cfg, seed, dev)
    x = torch.from_numpy(pkg.synth.make_input(n, r, seed)).to(dev)
    z = torch.from_numpy(pkg.synth.make_latent(n, cfg.z_dim, seed)).to(dev)
    cutoff = int(g["cutoff"]) if "cutoff" in g.files and int(g["cutoff"]) >= 0 else None
    with torch.no_grad():
        y = m(x, z=z, truncation_psi=float(g["psi"]), truncation_cutoff=cutoff, noise_mode="const")
        if cutoff is not None:                                                 # switching the cutoff on one module re-plans the workspace
            y_all = m(x, z=z, truncation_psi=float(g["psi"]), noise_mode="const")
            assert float((y - y_all).abs().max()) > 1e-2
            assert torch.equal(m(x, z=z, truncation_psi=float(g["psi"]), truncation_cutoff=cutoff, noise_mode="const"), y)
    err = float(np.abs(y.cpu().numpy() - g["y"]).max())
    assert np.isfinite(err) and err <= TOL, (tag, err, float(np.abs(g["y"]).max()))
    assert m._lib.backend() == "hip:gfx950"


def test_comodgan_512_batch_vs_oracle_and_properties(pkg, dev):
    """BASELINE.json configs[4] geometry (comodgan-512): parity of a batch of 2 against the CPU oracle; the same images inside a
    batch of 4 agree (images interact only through the batch-wide style normalisation, stylegan.py:139, which the
    demodulation cancels up to its 1e-8 epsilon); run-to-run determinism; input not modified."""
    cfg = _cfg(pkg, 512)
    m, sd = _build(pkg, cfg, 11, dev)
    x = pkg.synth.make_input(4, 512, 11)
    z = pkg.synth.make_latent(4, 512, 11)
    xt, zt = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
    x0 = xt.clone()
    with torch.no_grad():
        y2 = m(xt[:2], z=zt[:2], noise_mode="const").cpu().numpy()
        y4 = m(xt, z=zt, noise_mode="const").cpu().numpy()
        y4b = m(xt, z=zt, noise_mode="const").cpu().numpy()
    assert torch.equal(xt, x0)
    assert np.array_equal(y4, y4b)
    want = orc.generator(x[:2], z[:2], sd, 512, cfg.num_ws)
    scale = float(np.abs(want).max())
    err = float(np.abs(y2 - want).max())
    assert err <= TOL, (err, scale)
    assert float(np.abs(y4[:2] - y2).max()) <= 1e-4 * max(1.0, scale)


def test_noise_modes_and_truncation(pkg, dev):
    cfg = _cfg(pkg, 64, 4096, 64)
    m, sd = _build(pkg, cfg, 4, dev)
    n = 3
    x, z = pkg.synth.make_input(n, 64, 5), pkg.synth.make_latent(n, 512, 5)
    xt, zt = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
    with torch.no_grad():
        y_none = m(xt, z=zt, noise_mode="none", truncation_psi=0.5).cpu().numpy()
        torch.manual_seed(3)
        y_r1 = m(xt, z=zt, noise_mode="random").cpu().numpy()
        y_r2 = m(xt, z=zt, noise_mode="random").cpu().numpy()
        y_rz = m(xt, noise_mode="const")                       # z drawn inside (comodgan.py:438-439)
    want = orc.generator(x, z, sd, 64, cfg.num_ws, noise_mode="none", truncation_psi=0.5)
    assert float(np.abs(y_none - want).max()) <= TOL
    assert float(np.abs(y_r1 - y_r2).max()) > 1e-3             # fresh noise every call
    assert y_rz.shape == (n, 3, 64, 64) and bool(torch.isfinite(y_rz).all())
    # explicit noise through the C ABI against the oracle with the same draws
    h = m._engine(xt)
    per = h.noise_floats()
    noise = pkg.synth.normal((n * per,), 9, "drawn").astype(np.float32)
    nt = torch.from_numpy(noise).to(dev)
    ws = m._workspace(h, n, xt.device)
    y = torch.empty((n, 3, 64, 64), device=dev)
    h.forward(xt.data_ptr(), zt.data_ptr(), y.data_ptr(), n, ws.data_ptr(), ws.numel(), 1.0, "random", nt.data_ptr(),
              int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    want = orc.generator(x, z, sd, 64, cfg.num_ws, noise_mode="random", noise=noise)
    assert float(np.abs(y.cpu().numpy() - want).max()) <= TOL


def test_module_errors_on_gpu(pkg, dev):
    cfg = _cfg(pkg, 16, 1024, 64)
    m, _ = _build(pkg, cfg, 1, dev)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 16, 16), z=torch.zeros(1, 512))                    # CPU tensor
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 16, 16, device=dev))                               # wrong channel count
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 16, 16, device=dev), z=torch.zeros(2, 512, device=dev))
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 4, 16, 16, device=dev), return_intermediate_outs=True)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 4, 16, 16, device=dev), c=torch.zeros(1, 0, device=dev))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 16, 16, device=dev), truncation_cutoff=-2)


def test_in_place_weight_updates_are_seen_and_frozen_weights_are_reused(pkg, dev):
    """Default: the weight preparation runs every forward, so in-place updates (also through .data, which moves no version
    counter) always show up.  freeze_weights() is the explicit opt-in that prepares once: in-place writes are then NOT seen
    until freeze_weights() is called again; load_state_dict is picked up by itself.  Works on inference tensors."""
    cfg = _cfg(pkg, 32, 4096, 128)
    with torch.inference_mode():
        m, sd = _build(pkg, cfg, 8, dev)                              # parameters moved to the GPU as inference tensors
        x, z = pkg.synth.make_input(2, 32, 9), pkg.synth.make_latent(2, 512, 9)
        xt, zt = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
        y1 = m(xt, z=zt, noise_mode="const").cpu().numpy()
        y1b = m(xt, z=zt, noise_mode="const").cpu().numpy()
        m.encoder.b32.conv0.weight.data.mul_(1.5)                     # in place through .data: no version counter moves
        y2 = m(xt, z=zt, noise_mode="const").cpu().numpy()
        m.freeze_weights()
        y2b = m(xt, z=zt, noise_mode="const").cpu().numpy()          # prepares once more ...
        y2c = m(xt, z=zt, noise_mode="const").cpu().numpy()          # ... and reuses
        m.encoder.b32.conv0.weight.data.mul_(1.0 / 1.5)
        y2d = m(xt, z=zt, noise_mode="const").cpu().numpy()          # frozen: the write is not seen (the documented contract)
        m.freeze_weights()                                            # the documented way to invalidate
        y3 = m(xt, z=zt, noise_mode="const").cpu().numpy()
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})    # re-binding is picked up while frozen
        y4 = m(xt, z=zt, noise_mode="const").cpu().numpy()
    assert np.array_equal(y1, y1b) and np.array_equal(y2, y2b) and np.array_equal(y2, y2c) and np.array_equal(y2, y2d)
    sd2 = dict(sd)
    sd2["encoder.b32.conv0.weight"] = sd["encoder.b32.conv0.weight"] * np.float32(1.5)
    assert float(np.abs(y1 - orc.generator(x, z, sd, 32, cfg.num_ws)).max()) <= TOL
    assert float(np.abs(y2 - orc.generator(x, z, sd2, 32, cfg.num_ws)).max()) <= TOL
    assert float(np.abs(y2 - y1).max()) > 1e-2
    assert float(np.abs(y3 - y1).max()) <= 1e-4 * max(1.0, float(np.abs(y1).max()))
    assert float(np.abs(y4 - y1).max()) <= 1e-4 * max(1.0, float(np.abs(y1).max()))
