"""sepconv_wide2_kernel (mi-gan_amd/csrc/migan_wide2.hpp: persistent 16 x 16-pixel x 256-channel tiles for the 256 / 512-channel plain
layers) on the CPU fiber emulator, against the numpy oracle, through the C ABI entry migan_sepconv_forward.  The emulator defers every
LDS-DMA until the issuing lane's MIGAN_WAIT_VMCNT, so a mis-counted wait of the three rings (input tiles, weight planes, taps) leaves
NaN-poisoned LDS behind and fails these cases."""
import importlib

import numpy as np
import pytest

from tests.emu_util import emu_lib
from tests.sepconv_case import HostMem, run_sepconv_case

W2 = "migan::sepconv_wide2_kernel<"


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("mi-gan_amd")


# both forms of the weight-plane ring (tuning w2 = 1: 16-channel halves, 2: whole 32-channel chunks, the default)
@pytest.fixture(autouse=True, params=[1, 2])
def small_grids(request, lib):
    lib.set_tuning("w2", request.param)
    lib.set_tuning("w2_min_tiles", 1)
    lib.set_tuning("pipe_grid", 8)
    yield request.param
    lib.set_tuning("w2_min_tiles", 256)
    lib.set_tuning("pipe_grid", 256)
    lib.set_tuning("w2", 2)


# 8 workgroups: 1, 2 or 3 tiles each (first / steady-state / last tile), border and interior tiles, one and two column chunks
@pytest.mark.parametrize("cin,cout,h,w,batch", [(256, 256, 16, 16, 2), (256, 256, 32, 48, 2), (512, 512, 16, 32, 3), (64, 256, 48, 48, 1),
                                                (256, 512, 16, 16, 5)])
@pytest.mark.parametrize("noise", [False, True])
def test_plain_layers(lib, pkg, cin, cout, h, w, batch, noise):
    run_sepconv_case(lib, pkg, HostMem(), cin=cin, cout=cout, h=h, w=w, batch=batch, noise=noise, seed=23)
    assert lib.last_kernel().startswith(W2), lib.last_kernel()


def test_off_or_too_few_tiles_keeps_the_128_pixel_tile(lib, pkg, small_grids):
    lib.set_tuning("w2", 0)
    run_sepconv_case(lib, pkg, HostMem(), cin=256, cout=256, h=16, w=16, batch=2, noise=True, seed=23)
    assert lib.last_kernel().startswith("migan::sepconv_wide_kernel<"), lib.last_kernel()
    lib.set_tuning("w2", small_grids)
    lib.set_tuning("w2_min_tiles", 256)
    run_sepconv_case(lib, pkg, HostMem(), cin=256, cout=256, h=16, w=16, batch=2, noise=True, seed=23)
    assert lib.last_kernel().startswith("migan::sepconv_wide_kernel<"), lib.last_kernel()


def test_not_for_ragged_sizes_skip_or_torgb(lib, pkg):
    run_sepconv_case(lib, pkg, HostMem(), cin=256, cout=256, h=24, w=16, batch=2, seed=23)          # 24 rows: no whole 16 x 16 tiles
    assert lib.last_kernel().startswith("migan::sepconv_wide_kernel<"), lib.last_kernel()
    run_sepconv_case(lib, pkg, HostMem(), cin=256, cout=256, h=16, w=16, batch=2, skip=True, seed=23)
    assert lib.last_kernel().startswith("migan::sepconv_wide_kernel<"), lib.last_kernel()
    run_sepconv_case(lib, pkg, HostMem(), cin=256, cout=256, h=16, w=16, batch=2, noise=True, torgb=True, with_prev=True, seed=23)
    assert not lib.last_kernel().startswith(W2), lib.last_kernel()


@pytest.mark.parametrize("cin,cout,h,w", [(256, 256, 32, 32), (512, 512, 16, 16)])
def test_bit_identical_to_the_128_pixel_tile(lib, pkg, cin, cout, h, w):
    """same operand split, same order of the K chunks and of the three products: which tile form ran must not be visible in the result"""
    a = run_sepconv_case(lib, pkg, HostMem(), cin=cin, cout=cout, h=h, w=w, batch=2, noise=True, seed=29)
    assert lib.last_kernel().startswith(W2)
    lib.set_tuning("w2", 0)
    b = run_sepconv_case(lib, pkg, HostMem(), cin=cin, cout=cout, h=h, w=w, batch=2, noise=True, seed=29)
    assert lib.last_kernel().startswith("migan::sepconv_wide_kernel<")
    assert np.array_equal(a, b)


@pytest.mark.parametrize("cin,cout,h,w,batch", [(256, 512, 32, 32, 2), (512, 512, 32, 64, 3), (64, 256, 32, 32, 5)])
def test_pointwise_half_of_a_down2_layer(lib, pkg, cin, cout, h, w, batch):
    """down=2 through the two-kernel form (Cout = 512 has no fused kernel): dwfir_kernel, then the pointwise GEMM on the 256 x 256 tile"""
    lib.set_tuning("pipe", 7)                       # (no fused down=2 kernel: the 64 -> 256 case would otherwise take it)
    try:
        a = run_sepconv_case(lib, pkg, HostMem(), cin=cin, cout=cout, h=h, w=w, batch=batch, down=2, seed=37)
        assert lib.last_kernel() == "migan::sepconv_wide2_kernel<3>", lib.last_kernel()
        lib.set_tuning("w2_pw", 0)
        b = run_sepconv_case(lib, pkg, HostMem(), cin=cin, cout=cout, h=h, w=w, batch=batch, down=2, seed=37)
        assert lib.last_kernel().startswith("migan::sepconv_kernel<3,"), lib.last_kernel()
        assert np.array_equal(a, b)
    finally:
        lib.set_tuning("pipe", 15)
        lib.set_tuning("w2_pw", 1)
