"""f16x2 split GEMM against adversarial intra-tensor operand ranges on the MI355X (tests/robust_case.py), plus the non-finite-input
policy measured against what the reference module (its torch-CPU port, bit-exact with it) does with the same input."""
import numpy as np
import pytest
import torch

from oracle import migan_torch_cpu as torc
from tests.robust_case import KINDS, check_kind
from tests.sepconv_case import CudaMem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("gpu tests need an MI355X (torch.cuda.is_available() is False)")
    return torch.device("cuda", 0)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("geo", [dict(cin=64, cout=64, h=32, batch=2), dict(cin=128, cout=256, h=16, batch=1), dict(cin=512, cout=512, h=8, batch=2)])
def test_f16x2_matches_exact_fp32_mfma_on_adversarial_operands(pkg, dev, kind, geo):
    e32, e16, ymax = check_kind(pkg.load_library(), pkg, CudaMem(dev), kind, **geo)
    print(f"{kind} {geo}: |y|max {ymax:.3e} err f32 {e32:.3e} f16x2 {e16:.3e}")


def test_nan_input_reference_propagates_kernels_clamp(pkg, dev):
    """Documented divergence (behavioural differences): `Tensor.clamp` (reference lrelu_agc :21-23) propagates a NaN,
    so ONE NaN input pixel turns the reference module's whole output into NaN (the 4x4 bottleneck mixes every pixel into every
    other); the kernels' clamp is v_med3_f32, which returns -256 for a NaN activation -- the behaviour of the reference's own CUDA
    plugin (torch_utils/ops/bias_act.cu:139) -- so the damage stays finite.  Non-finite inputs are outside the path's contract
    (images in [-1, 1], masks in {0, 1}); the test pins both behaviours so neither changes silently."""
    res = 64
    sd = pkg.synth.make_state_dict(res, seed=61, regime="export")
    m = pkg.Generator(resolution=res)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to(dev).eval()
    x = pkg.synth.make_input(1, res, seed=61)
    x[0, 1, 10, 10] = np.nan
    ref = torc.generator(x, sd, res).numpy()
    assert np.isnan(ref).all()                                   # the reference: everything is lost
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.isfinite(y).all()                                  # the kernels: finite everywhere
    clean = x.copy()
    clean[0, 1, 10, 10] = 0.0
    with torch.no_grad():
        yc = m(torch.from_numpy(clean).to(dev)).cpu().numpy()
    # ... and far from the bad pixel only the globally mixed part of the signal moved
    assert float(np.abs(y - yc)[..., 40:, 40:].max()) < float(np.abs(yc).max())


# ---- the NaN-propagating build (libmigan_hip_strictnan.so, -DMIGAN_STRICT_NAN): Tensor.clamp's behaviour, reference :21-23 -------------
@pytest.fixture(scope="module")
def strict_lib(pkg):
    lib = pkg.hipbind.load_library(nan_policy="propagate")
    assert lib.nan_policy() == "propagate" and lib.backend() == "hip:gfx950"
    return lib


@pytest.mark.parametrize("kw", [dict(cin=64, cout=64, h=32, batch=2), dict(cin=64, cout=128, h=16, batch=2, down=2),
                                dict(cin=128, cout=64, h=16, batch=2, up=2, noise=True, skip=True), dict(cin=256, cout=256, h=16, batch=1)])
def test_strict_nan_operator_mask_follows_the_oracle(pkg, dev, strict_lib, kw):
    """one NaN input element: the output is NaN exactly where the oracle's is (the 3x3 neighbourhood, every output channel, FIR spread)"""
    from tests.sepconv_case import run_sepconv_case
    strict_lib.set_tuning("pipe_min_tiles", 1)
    try:
        run_sepconv_case(strict_lib, pkg, CudaMem(dev), seed=13, nan_at=(0, 5, 7, 9), **kw)
    finally:
        strict_lib.set_tuning("pipe_min_tiles", 256)


def test_strict_nan_generator_matches_the_reference_and_the_default_build(pkg, dev):
    res = 64
    sd = pkg.synth.make_state_dict(res, seed=61, regime="export")

    def model(policy):
        m = pkg.Generator(resolution=res, nan_policy=policy)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        return m.to(dev).eval()

    strict, default = model("propagate"), model("clamp")
    x = pkg.synth.make_input(2, res, seed=61)
    with torch.no_grad():
        ys, yd = strict(torch.from_numpy(x).to(dev)).cpu().numpy(), default(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(ys, yd)                                 # finite inputs: the two builds compute the same bits
    x[0, 1, 10, 10] = np.nan                                      # image 0 only
    ref = torc.generator(x, sd, res).numpy()
    with torch.no_grad():
        y = strict(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(np.isnan(y), np.isnan(ref))             # Tensor.clamp semantics: image 0 is lost, image 1 untouched
    assert np.isnan(ref[0]).all() and np.isfinite(ref[1]).all()
    np.testing.assert_allclose(y[1], ref[1], rtol=0, atol=1e-4 * max(1.0, float(np.abs(ref[1]).max())))
