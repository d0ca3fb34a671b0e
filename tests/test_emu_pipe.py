"""The software-pipelined persistent SeparableConv2d kernels (mi-gan_amd/csrc/migan_pipe.hpp) on the CPU fiber emulator, against the
numpy oracle, through the C ABI entry migan_sepconv_forward.  The emulator defers every LDS-DMA until the issuing lane's
MIGAN_WAIT_VMCNT, so a missing or mis-counted wait of the DMA ring leaves NaN-poisoned LDS behind and fails these cases; its
lanes run to the next collective one after the other, so a missing barrier between the wave groups gives a wrong result too."""
import importlib

import numpy as np
import pytest

from tests.emu_util import emu_lib
from tests.sepconv_case import HostMem, run_sepconv_case


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("mi-gan_amd")


# both workgroup shapes: 4 or 8 waves in the depthwise group (12- / 16-wave workgroups)
@pytest.fixture(autouse=True, params=[4, 8])
def small_grids(request, lib):
    # the emulator cases have a few dozen tiles: let them take the pipelined kernels, a few tiles per workgroup
    lib.set_tuning("pipe_min_tiles", 1)
    lib.set_tuning("pipe_grid", 8)
    lib.set_tuning("pipe_na", request.param)
    lib.set_tuning("pipe_na8", 0)      # (this wave count for every form)
    yield request.param
    lib.set_tuning("pipe_dna", 12)
    lib.set_tuning("pipe_min_tiles", 256)
    lib.set_tuning("pipe_grid", 256)
    lib.set_tuning("pipe", 15)
    lib.set_tuning("pipe_na", 4)
    lib.set_tuning("pipe_na8", 9)


PIPE = "migan::sepconv_pipe_kernel<"


# (h, w, batch): 8x16 tiles; with 8 workgroups some walk 1 tile, some 2, some 3 (steady state, first and last tile)
@pytest.mark.parametrize("h,w,batch", [(16, 32, 2), (16, 32, 5), (8, 16, 2), (24, 48, 2)])
@pytest.mark.parametrize("noise", [False, True])
def test_plain_64_to_64(lib, pkg, h, w, batch, noise):
    run_sepconv_case(lib, pkg, HostMem(), cin=64, cout=64, h=h, w=w, batch=batch, noise=noise, seed=3)
    assert lib.last_kernel().startswith(PIPE + "0, 64, 64, false, false"), lib.last_kernel()


@pytest.mark.parametrize("h,w,batch,prev", [(16, 32, 3, True), (16, 16, 2, False), (32, 32, 2, True)])
def test_plain_with_fused_torgb(lib, pkg, h, w, batch, prev):
    run_sepconv_case(lib, pkg, HostMem(), cin=64, cout=64, h=h, w=w, batch=batch, noise=True, torgb=True, with_prev=prev, seed=5)
    assert lib.last_kernel().startswith(PIPE + "0, 64, 64, false, true"), lib.last_kernel()


@pytest.mark.parametrize("h,w,batch", [(16, 32, 3), (8, 16, 2), (32, 16, 2)])
def test_plain_with_fused_fromrgb(lib, pkg, h, w, batch):
    run_sepconv_case(lib, pkg, HostMem(), cin=64, cout=64, h=h, w=w, batch=batch, fromrgb=True, seed=7)
    assert lib.last_kernel().startswith(PIPE + "0, 64, 64, true, false"), lib.last_kernel()


@pytest.mark.parametrize("h,w,batch", [(8, 16, 2), (16, 32, 2), (12, 20, 3), (6, 14, 2)])
@pytest.mark.parametrize("noise,skip", [(True, True), (False, False)])
def test_fir_up_128_to_64(lib, pkg, h, w, batch, noise, skip):
    run_sepconv_case(lib, pkg, HostMem(), cin=128, cout=64, h=h, w=w, batch=batch, up=2, noise=noise, skip=skip, seed=9)
    assert lib.last_kernel().startswith(PIPE + "2, 64, 128, false, false"), lib.last_kernel()


def test_plain_layer_with_a_skip_tensor_keeps_the_one_tile_kernel(lib, pkg):
    """the pipelined plain epilogue has no skip add: such a call through the operator ABI must not take it"""
    run_sepconv_case(lib, pkg, HostMem(), cin=64, cout=64, h=16, w=32, batch=2, noise=True, skip=True, seed=3)
    assert lib.last_kernel().startswith("migan::sepconv_kernel<"), lib.last_kernel()
    run_sepconv_case(lib, pkg, HostMem(), cin=128, cout=128, h=16, w=32, batch=2, noise=True, skip=True, seed=3)
    assert lib.last_kernel().startswith("migan::sepconv_kernel<"), lib.last_kernel()


def test_pipe_off_takes_the_one_tile_kernels(lib, pkg):
    lib.set_tuning("pipe", 0)
    run_sepconv_case(lib, pkg, HostMem(), cin=64, cout=64, h=16, w=32, batch=2, noise=True, seed=3)
    assert lib.last_kernel().startswith("migan::sepconv_kernel<"), lib.last_kernel()


def test_single_image_forwards_take_them_too_unless_told_otherwise(lib, pkg):
    run_sepconv_case(lib, pkg, HostMem(), cin=64, cout=64, h=16, w=32, batch=1, noise=True, seed=3)
    assert lib.last_kernel().startswith(PIPE), lib.last_kernel()
    lib.set_tuning("pipe_min_batch", 2)
    try:
        run_sepconv_case(lib, pkg, HostMem(), cin=64, cout=64, h=16, w=32, batch=1, noise=True, seed=3)
        assert lib.last_kernel().startswith("migan::sepconv_kernel<"), lib.last_kernel()
    finally:
        lib.set_tuning("pipe_min_batch", 1)


DOWN = "migan::sepconv_pipedown_kernel<"


@pytest.mark.parametrize("h,w,batch", [(16, 32, 2), (8, 32, 3), (24, 64, 2), (32, 32, 5)])
@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 256)])
@pytest.mark.parametrize("dna,nb", [(4, 8), (8, 8), (12, 4)])
def test_fused_down(lib, pkg, small_grids, h, w, batch, cin, cout, dna, nb):
    """down=2 as one launch: depthwise + FIR-down feed the 1x1 through LDS (4 x 16 low-resolution tiles, image borders = the FIR's zero padding);
    4, 8 or 12 depthwise + FIR waves beside 8, 8 or 4 GEMM / epilogue waves (with 4 and 256 columns: the epilogue at the tile's end)"""
    if small_grids != 4:
        pytest.skip("this kernel's wave split is its own parameter")
    lib.set_tuning("pipe", 15)
    lib.set_tuning("pipe_dna", dna)
    run_sepconv_case(lib, pkg, HostMem(), cin=cin, cout=cout, h=h, w=w, batch=batch, down=2, seed=11)
    if (cin, dna) == (128, 8):
        dna = 4                                         # (no 8 + 8 instantiation for 128 -> 256: two registers short)
    assert lib.last_kernel() == DOWN + f"{cout}, {cin}, 2, {dna}, {nb}>", lib.last_kernel()


def test_fused_down_off_takes_the_two_kernel_form(lib, pkg):
    lib.set_tuning("pipe", 7)
    run_sepconv_case(lib, pkg, HostMem(), cin=64, cout=128, h=16, w=32, batch=2, down=2, seed=11)
    assert lib.last_kernel().startswith("migan::sepconv_kernel<3,"), lib.last_kernel()


@pytest.mark.parametrize("cin,cout,h,w,batch", [(64, 128, 16, 32, 2), (128, 256, 24, 64, 2)])
def test_fused_down_equals_the_two_kernel_form_bit_for_bit(lib, pkg, cin, cout, h, w, batch):
    """since round 5 both forms of a down=2 layer evaluate the [1,3,3,1]^2 / 64 FIR separably (vertical (1,3)/8 and (3,1)/8
    partial sums, then the horizontal half) -- another rounding than the 16-tap sum of rounds 1-4, and the same one in both kernels.  The
    fused launch (sepconv_pipedown_kernel) and dwfir_kernel + pointwise GEMM must therefore stay in lockstep: same bits."""
    lib.set_tuning("pipe", 15)
    a = run_sepconv_case(lib, pkg, HostMem(), cin=cin, cout=cout, h=h, w=w, batch=batch, down=2, seed=13)
    assert lib.last_kernel().startswith(DOWN), lib.last_kernel()
    lib.set_tuning("pipe", 7)
    try:
        b = run_sepconv_case(lib, pkg, HostMem(), cin=cin, cout=cout, h=h, w=w, batch=batch, down=2, seed=13)
        assert lib.last_kernel().startswith("migan::sepconv_kernel<3,"), lib.last_kernel()
    finally:
        lib.set_tuning("pipe", 15)
    assert np.array_equal(a, b)

@pytest.mark.parametrize("h,w,batch", [(16, 32, 2), (8, 16, 3), (24, 16, 2)])
@pytest.mark.parametrize("torgb", [False, True])
def test_plain_128_to_128(lib, pkg, h, w, batch, torgb):
    """the 256 x 256 layers of migan-512: 128-column tiles (two blocks per B wave), weight planes streamed through the two-slot ring"""
    run_sepconv_case(lib, pkg, HostMem(), cin=128, cout=128, h=h, w=w, batch=batch, noise=True, torgb=torgb, with_prev=torgb, seed=19)
    assert lib.last_kernel().startswith(PIPE + "0, 128, 128, false, " + ("true" if torgb else "false")), lib.last_kernel()
