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
