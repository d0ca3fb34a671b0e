"""SeparableConv2d with fewer than 64 channels (the layers of generators above 512: channels(1024) = 32, reference :222-223) -- the plain
kernel `narrow_sepconv_kernel` of the product source on the CPU emulator, through migan_sepconv_forward, against the numpy oracle."""
import importlib

import pytest

from tests.emu_util import emu_lib
from tests.sepconv_case import HostMem, run_sepconv_case

NARROW = "migan::narrow_sepconv_kernel<"


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("mi-gan_amd")


CASES = [
    (dict(cin=32, cout=32, h=12, w=20, batch=2, noise=True, skip=True), "0, false>"),          # synthesis.b1024.conv2-like, ragged size
    (dict(cin=32, cout=32, h=8, batch=1), "0, false>"),
    (dict(cin=16, cout=64, h=16, w=24, batch=2, down=2), "1, false>"),                          # FIR-down (encoder.b1024.conv2, 32 -> 64, is wide enough for the tiled kernels)
    (dict(cin=16, cout=32, h=8, batch=3, down=2), "1, false>"),                                  # encoder.b2048.conv2
    (dict(cin=64, cout=32, h=6, w=10, batch=2, up=2, noise=True, skip=True), "2, false>"),     # synthesis.b1024.conv1: 64 -> 32, FIR-up
    (dict(cin=16, cout=8, h=4, batch=2, up=2, noise=True), "2, false>"),
    (dict(cin=32, cout=32, h=16, batch=2, fromrgb=True), "0, true>"),                            # encoder.b1024.conv1 with the fused FromRGB
    (dict(cin=32, cout=32, h=16, batch=2, noise=True, torgb=True, with_prev=True), "0, false>"),   # ... and the ToRGB tail
    (dict(cin=8, cout=8, h=8, batch=1, noise=True, torgb=True), "0, false>"),
]


@pytest.mark.parametrize("kw,kernel", CASES)
def test_narrow_layers(lib, pkg, kw, kernel):
    run_sepconv_case(lib, pkg, HostMem(), seed=41, **kw)
    assert lib.last_kernel() == NARROW + kernel, lib.last_kernel()


@pytest.mark.parametrize("kw", [dict(cin=32, cout=32, h=16, batch=2, noise=True), dict(cin=16, cout=32, h=16, batch=2, down=2),
                                dict(cin=64, cout=32, h=16, batch=1, up=2, noise=True, skip=True)])
def test_narrow_layers_propagate_a_nan_like_the_oracle(lib, pkg, kw):
    run_sepconv_case(lib, pkg, HostMem(), seed=13, nan_at=(0, 5, 7, 9), **kw)


def test_16_bit_storage_is_refused(lib, pkg):
    with pytest.raises(NotImplementedError):
        run_sepconv_case(lib, pkg, HostMem(), cin=32, cout=32, h=8, batch=1, storage="bf16", seed=1)
