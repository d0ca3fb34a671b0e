"""f16x2 split GEMM against adversarial intra-tensor operand ranges, on the CPU emulator of the product kernel source
(tests/robust_case.py; the GPU leg is tests/test_gpu_robust.py)."""
import importlib

import pytest

from tests.emu_util import emu_lib
from tests.robust_case import KINDS, check_kind
from tests.sepconv_case import HostMem


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.mark.parametrize("kind", KINDS)
def test_f16x2_matches_exact_fp32_mfma_on_adversarial_operands(pkg, lib, kind):
    check_kind(lib, pkg, HostMem(), kind, h=8)


# ---- NaN policy of the default build: a NaN stays a NaN through lrelu_agc's clamp, like Tensor.clamp (reference :21-23) --------------------
NAN_CASES = [dict(cin=64, cout=64, h=16, batch=2), dict(cin=64, cout=128, h=16, w=32, batch=2, down=2),
             dict(cin=128, cout=64, h=8, w=16, batch=2, up=2, noise=True, skip=True), dict(cin=256, cout=256, h=16, batch=1, noise=True),
             dict(cin=64, cout=64, h=16, batch=2, noise=True, torgb=True, with_prev=True), dict(cin=128, cout=256, h=16, w=32, batch=1, down=2)]


@pytest.mark.parametrize("min_tiles", [256, 1])          # 1: the pipelined / 256-pixel-tile kernels take these small cases
@pytest.mark.parametrize("kw", NAN_CASES)
def test_nan_mask_follows_the_oracle(pkg, lib, kw, min_tiles):
    """one NaN input element: the product kernel source (clamp4 / clamp1: pairwise unordered compare + repair) leaves NaNs exactly where the
    oracle has them -- the 3x3 neighbourhood of the element, every output channel, spread by the FIR of the down / up layers"""
    from tests.sepconv_case import run_sepconv_case
    assert lib.nan_policy() == "propagate"
    lib.set_tuning("pipe_min_tiles", min_tiles)
    lib.set_tuning("w2_min_tiles", min_tiles)
    lib.set_tuning("pipe_grid", 8 if min_tiles == 1 else 256)
    try:
        run_sepconv_case(lib, pkg, HostMem(), seed=13, nan_at=(0, 5, 7, 9), **kw)
    finally:
        lib.set_tuning("pipe_min_tiles", 256)
        lib.set_tuning("w2_min_tiles", 256)
        lib.set_tuning("pipe_grid", 256)
