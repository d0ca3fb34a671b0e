"""SeparableConv2d with fewer than 64 channels (the layers of generators above 512: channels(1024) = 32, reference :222-223) -- the plain
kernel `narrow_sepconv_kernel` on the MI355X, through migan_sepconv_forward, against the numpy oracle; and a Generator(1024) forward on
an any-size input, against the torch-CPU port of the reference."""
import importlib

import pytest

import numpy as np
import torch

from oracle import migan_torch_cpu as torc
from tests.sepconv_case import CudaMem, run_sepconv_case

pytestmark = pytest.mark.gpu

NARROW = "migan::narrow_sepconv_kernel<"


@pytest.fixture(scope="module")
def lib(pkg):
    return pkg.load_library()


@pytest.fixture(scope="module")
def mem():
    if not torch.cuda.is_available():
        pytest.skip("gpu tests need an MI355X (torch.cuda.is_available() is False)")
    return CudaMem(torch.device("cuda", 0))


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
def test_narrow_layers(lib, pkg, mem, kw, kernel):
    run_sepconv_case(lib, pkg, mem, seed=41, **kw)
    assert lib.last_kernel() == NARROW + kernel, lib.last_kernel()


@pytest.mark.parametrize("kw", [dict(cin=32, cout=32, h=16, batch=2, noise=True), dict(cin=16, cout=32, h=16, batch=2, down=2),
                                dict(cin=64, cout=32, h=16, batch=1, up=2, noise=True, skip=True)])
def test_narrow_layers_propagate_a_nan_like_the_oracle(lib, pkg, mem, kw):
    run_sepconv_case(lib, pkg, mem, seed=13, nan_at=(0, 5, 7, 9), **kw)


def test_16_bit_storage_is_refused(lib, pkg, mem):
    with pytest.raises(NotImplementedError):
        run_sepconv_case(lib, pkg, mem, cin=32, cout=32, h=8, batch=1, storage="bf16", seed=1)


def test_generator_1024_any_size_and_batch(pkg, mem):
    """Generator(1024): a 256 x 512 input (multiples of R / 4) through forward_any_size, batch 3 -- every block of the 1024 model at a
    quarter / half of its size, the 32-channel blocks on the plain kernel, the rest on the tiled kernels -- against the torch-CPU port of
    the reference (which agrees with the reference module on the committed 1024 x 1024 golden)."""
    dev = torch.device("cuda", 0)
    res, seed = 1024, 23
    sd = pkg.synth.make_state_dict(res, seed=seed, regime="export")
    m = pkg.Generator(resolution=res)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to(dev).eval()
    x = (pkg.synth.normal((3, 4, 256, 512), seed, "xhw") * 0.7).astype(np.float32)
    with torch.no_grad():
        y = m.forward_any_size(torch.from_numpy(x).to(dev)).cpu().numpy()
    want = torc.generator(x, sd, res).numpy()
    assert np.abs(y - want).max() <= 1e-3, float(np.abs(y - want).max())
    kernels = [l["kernel"] for l in m.launch_info()]
    assert sum(k.startswith(NARROW) for k in kernels) == 3          # encoder.b1024.conv1, synthesis.b1024.conv1 / conv2 (encoder.b1024.conv2, 32 -> 64, runs the tiled kernels)
    with pytest.raises((NotImplementedError, RuntimeError)):        # the uint8 forward is not offered above 512
        m.forward_uint8(torch.zeros((1, res, res, 3), dtype=torch.uint8, device=dev), torch.zeros((1, res, res), dtype=torch.uint8, device=dev))
