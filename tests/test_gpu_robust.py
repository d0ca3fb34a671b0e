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


def test_nan_input_default_follows_the_reference_and_the_clamp_build_stays_finite(pkg, dev):
    """`Tensor.clamp` (reference lrelu_agc :21-23) propagates a NaN, so ONE NaN input pixel turns the reference module's whole output into
    NaN (the 4x4 bottleneck mixes every pixel into every other).  Since round 6 the DEFAULT library does the same ("follow
    torch"); the opt-in `nan_policy="clamp"` build keeps rounds 1-5's behaviour -- v_med3_f32 returns -256 for a NaN activation, like the
    reference's own CUDA plugin (torch_utils/ops/bias_act.cu:139) -- so there the damage stays finite.  The test pins both."""
    res = 64
    sd = pkg.synth.make_state_dict(res, seed=61, regime="export")

    def model(policy):
        m = pkg.Generator(resolution=res, nan_policy=policy)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
        return m.to(dev).eval()

    default, clamp = model("propagate"), model("clamp")
    assert pkg.Generator(resolution=res)._nan_policy == "propagate"
    assert default._engine(torch.zeros(1, 4, res, res, device=dev)) is not None and default._lib.nan_policy() == "propagate"
    x = pkg.synth.make_input(1, res, seed=61)
    x[0, 1, 10, 10] = np.nan
    ref = torc.generator(x, sd, res).numpy()
    assert np.isnan(ref).all()                                   # the reference: everything is lost
    with torch.no_grad():
        y = default(torch.from_numpy(x).to(dev)).cpu().numpy()
        yk = clamp(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(np.isnan(y), np.isnan(ref))            # the default build: the reference's mask
    assert clamp._lib.nan_policy() == "clamp" and np.isfinite(yk).all()      # the opt-in build: finite everywhere
    clean = x.copy()
    clean[0, 1, 10, 10] = 0.0
    with torch.no_grad():
        yc = clamp(torch.from_numpy(clean).to(dev)).cpu().numpy()
        yd = default(torch.from_numpy(clean).to(dev)).cpu().numpy()
    # ... and far from the bad pixel only the globally mixed part of the signal moved
    assert float(np.abs(yk - yc)[..., 40:, 40:].max()) < float(np.abs(yc).max())
    # finite inputs: the two builds give the same bits
    assert np.array_equal(yc, yd)


# ---- NaN masks of the default library against the oracle, operator by operator and through every kernel family -------------------------
@pytest.fixture(scope="module")
def default_lib(pkg):
    lib = pkg.hipbind.load_library()
    assert lib.nan_policy() == "propagate" and lib.backend() == "hip:gfx950"
    return lib


NAN_CASES = [dict(cin=64, cout=64, h=32, batch=2), dict(cin=64, cout=128, h=16, batch=2, down=2),
             dict(cin=128, cout=64, h=16, batch=2, up=2, noise=True, skip=True), dict(cin=256, cout=256, h=16, batch=1),
             dict(cin=64, cout=64, h=16, batch=2, noise=True, torgb=True, with_prev=True), dict(cin=256, cout=256, h=32, batch=4, noise=True),
             dict(cin=512, cout=512, h=32, batch=2), dict(cin=64, cout=128, h=64, batch=4, down=2), dict(cin=256, cout=512, h=32, batch=2, down=2),
             dict(cin=128, cout=128, h=32, batch=2, noise=True, torgb=True, with_prev=True), dict(cin=512, cout=256, h=16, batch=2, up=2, noise=True, skip=True)]


@pytest.mark.parametrize("min_tiles", [256, 1])          # 1: the pipelined / 256-pixel-tile kernels take the small cases too
@pytest.mark.parametrize("kw", NAN_CASES)
def test_nan_operator_mask_follows_the_oracle(pkg, dev, default_lib, kw, min_tiles):
    """one NaN input element: the output is NaN exactly where the oracle's is (the 3x3 neighbourhood, every output channel, FIR spread)"""
    from tests.sepconv_case import run_sepconv_case
    default_lib.set_tuning("pipe_min_tiles", min_tiles)
    default_lib.set_tuning("w2_min_tiles", min_tiles)
    try:
        run_sepconv_case(default_lib, pkg, CudaMem(dev), seed=13, nan_at=(0, 5, 7, 9), **kw)
    finally:
        default_lib.set_tuning("pipe_min_tiles", 256)
        default_lib.set_tuning("w2_min_tiles", 256)


def test_nan_generator_mask_matches_the_reference_with_several_bad_pixels(pkg, dev):
    """a NaN that cannot reach the whole image (resolution 8 ... the mask is still everything); and a batch where only ONE image is bad:
    the other images must be untouched, bit for bit"""
    res = 64
    sd = pkg.synth.make_state_dict(res, seed=62, regime="export")
    m = pkg.Generator(resolution=res)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to(dev).eval()
    x = pkg.synth.make_input(4, res, seed=62)
    with torch.no_grad():
        clean = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    x[2, 0, 3, 60] = np.nan
    ref = torc.generator(x, sd, res).numpy()
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(np.isnan(y), np.isnan(ref)) and np.isnan(y[2]).all()
    assert np.array_equal(y[[0, 1, 3]], clean[[0, 1, 3]])
