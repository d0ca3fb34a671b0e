"""Real image / mask pairs of the reference's examples/places2_512_object, pushed through the reference's own demo.py functions
and Generator by tests/golden/make_golden_examples.py (non-square originals, object masks).  CPU: the oracles and the
pre/post-processing restatement against those fixtures.  GPU (-m gpu): pipeline + HIP forward + the fused uint8 forward."""
import os

import numpy as np
import pytest
import torch

from oracle import migan_prepost as pp
from oracle import migan_torch_cpu as torc

CASES = ["p256", "p512"]


def _load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, f"examples_{tag}.npz"))


@pytest.mark.parametrize("tag", CASES)
def test_oracle_on_real_examples(pkg, golden_dir, tag):
    g = _load(golden_dir, tag)
    res, seed = int(g["resolution"]), int(g["seed"])
    assert any(w != h for w, h in g["orig_sizes"])                       # non-square originals went through demo.resize
    x = pp.preprocess(g["img_u8"], g["mask_u8"])
    np.testing.assert_allclose(x.astype(np.float64).sum(axis=(2, 3)), g["x_sum"], rtol=0, atol=1e-6 * res * res)
    np.testing.assert_allclose(np.abs(x).astype(np.float64).sum(axis=(2, 3)), g["x_abs_sum"], rtol=0, atol=1e-6 * res * res)
    sd = pkg.synth.make_state_dict(res, seed=seed, regime="export")
    y = torc.generator(x, sd, res).numpy()
    s = int(g["stride"])
    tol = 3e-5 * max(1.0, float(g["y_absmax"]))
    np.testing.assert_allclose(y[:, :, ::s, ::s], g["y"], rtol=0, atol=tol)
    np.testing.assert_allclose(y.astype(np.float64).sum(axis=(2, 3)), g["y_sum"], rtol=0, atol=tol * res * res)
    comp = pp.compose(y, g["img_u8"], g["mask_u8"])
    d = np.abs(comp.astype(np.int32) - g["composed"].astype(np.int32))
    assert d.max() <= 1 and float((d > 0).mean()) < 1e-3                # uint8 truncation of values 1e-5 apart


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_forward_on_real_examples(pkg, golden_dir, tag):
    if not torch.cuda.is_available():
        pytest.skip("gpu tests need an MI355X (torch.cuda.is_available() is False)")
    dev = torch.device("cuda", 0)
    g = _load(golden_dir, tag)
    res, seed = int(g["resolution"]), int(g["seed"])
    sd = pkg.synth.make_state_dict(res, seed=seed, regime="export")
    m = pkg.Generator(resolution=res)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to(dev).eval()
    img, mask = torch.from_numpy(g["img_u8"]).to(dev), torch.from_numpy(g["mask_u8"]).to(dev)
    with torch.no_grad():
        x = pkg.pipeline.preprocess(img, mask)
        y = m(x)
        comp = pkg.pipeline.compose(y, img, mask)
        comp_fused = m.forward_uint8(img, mask)
    np.testing.assert_allclose(x.double().sum(dim=(2, 3)).cpu().numpy(), g["x_sum"], rtol=0, atol=1e-6 * res * res)
    s = int(g["stride"])
    yn = y.cpu().numpy()
    assert float(np.abs(yn[:, :, ::s, ::s] - g["y"]).max()) <= 1e-4 * max(1.0, float(g["y_absmax"]))
    assert float(np.abs(yn[:, :, ::s, ::s] - g["y"]).max()) <= 1e-3 * max(1.0, float(g["y_absmax"]) / 30.0)
    assert torch.equal(comp, comp_fused)
    d = np.abs(comp.cpu().numpy().astype(np.int32) - g["composed"].astype(np.int32))
    assert d.max() <= 1 and float((d > 0).mean()) < 1e-3
