"""sepconv_wide2_kernel (persistent 16 x 16-pixel x 256-channel tiles, mi-gan_amd/csrc/migan_wide2.hpp) on the MI355X, through the C ABI
entry migan_sepconv_forward: against the numpy oracle at the layer sizes of migan-512 / migan-256, and bit for bit against the
128-pixel tile kernel it replaces (same operand split, same summation order)."""
import numpy as np
import pytest

from tests.sepconv_case import CudaMem, run_sepconv_case

pytestmark = pytest.mark.gpu
W2 = "migan::sepconv_wide2_kernel<"


@pytest.fixture(scope="module")
def lib(pkg):
    return pkg.load_library()


@pytest.fixture(scope="module")
def mem():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("gpu tests need an MI355X (torch.cuda.is_available() is False)")
    return CudaMem(torch.device("cuda", 0))


# both forms of the weight-plane ring (tuning w2 = 1: 16-channel halves, 2: whole 32-channel chunks, the default)
@pytest.fixture(autouse=True, params=[1, 2])
def restore(request, lib):
    default = 2
    lib.set_tuning("w2", request.param)
    yield
    lib.set_tuning("w2", default)
    lib.set_tuning("w2_min_tiles", 256)
    lib.set_tuning("pipe_grid", 256)


# (cin, cout, h, w, batch): encoder.b128.conv1 / synthesis.b64.conv2 / encoder.b32.conv1 of migan-512 at batch 4..32, a non-square size,
# a launch with fewer tiles than workgroups and one where the workgroups walk 1..3 tiles each
@pytest.mark.parametrize("cin,cout,h,w,batch", [(256, 256, 128, 128, 4), (512, 512, 64, 64, 8), (512, 512, 32, 32, 32), (256, 512, 48, 80, 3),
                                                (256, 256, 64, 64, 9)])
@pytest.mark.parametrize("noise", [False, True])
def test_plain_layers(lib, pkg, mem, cin, cout, h, w, batch, noise):
    lib.set_tuning("w2_min_tiles", 1)
    run_sepconv_case(lib, pkg, mem, cin=cin, cout=cout, h=h, w=w, batch=batch, noise=noise, seed=31)
    assert lib.last_kernel().startswith(W2), lib.last_kernel()


@pytest.mark.parametrize("grid", [8, 64, 256])
def test_any_number_of_tiles_per_workgroup(lib, pkg, mem, grid):
    lib.set_tuning("w2_min_tiles", 1)
    lib.set_tuning("pipe_grid", grid)
    run_sepconv_case(lib, pkg, mem, cin=256, cout=256, h=64, w=64, batch=5, noise=True, seed=33)
    assert lib.last_kernel().startswith(W2), lib.last_kernel()


@pytest.mark.parametrize("noise", [True, False])          # (False: the act1g epilogue without the noise term -- ADVICE round 5)
@pytest.mark.parametrize("cin,cout,h,batch", [(256, 256, 128, 8), (512, 512, 64, 16)])
def test_bit_identical_to_the_128_pixel_tile_and_run_to_run(lib, pkg, mem, cin, cout, h, batch, noise):
    a = run_sepconv_case(lib, pkg, mem, cin=cin, cout=cout, h=h, w=h, batch=batch, noise=noise, seed=35)
    assert lib.last_kernel().startswith(W2), lib.last_kernel()
    a2 = run_sepconv_case(lib, pkg, mem, cin=cin, cout=cout, h=h, w=h, batch=batch, noise=noise, seed=35)
    assert np.array_equal(a, a2)
    lib.set_tuning("w2", 0)
    b = run_sepconv_case(lib, pkg, mem, cin=cin, cout=cout, h=h, w=h, batch=batch, noise=noise, seed=35)
    assert lib.last_kernel().startswith("migan::sepconv_wide_kernel<"), lib.last_kernel()
    assert np.array_equal(a, b)


@pytest.mark.parametrize("cin,cout,h,batch", [(256, 512, 128, 8), (512, 512, 64, 32)])
def test_pointwise_half_of_a_down2_layer(lib, pkg, mem, cin, cout, h, batch):
    """encoder.b128.conv2 / b64.conv2 of migan-512 (down=2, Cout = 512: dwfir_kernel + pointwise GEMM): the 256 x 256 tile against the oracle
    and bit for bit against the 128-column pointwise tiles"""
    a = run_sepconv_case(lib, pkg, mem, cin=cin, cout=cout, h=h, w=h, batch=batch, down=2, seed=39)
    assert lib.last_kernel() == "migan::sepconv_wide2_kernel<3>", lib.last_kernel()
    lib.set_tuning("w2_pw", 0)
    try:
        b = run_sepconv_case(lib, pkg, mem, cin=cin, cout=cout, h=h, w=h, batch=batch, down=2, seed=39)
        assert lib.last_kernel().startswith("migan::sepconv_kernel<3,"), lib.last_kernel()
    finally:
        lib.set_tuning("w2_pw", 1)
    assert np.array_equal(a, b)


def test_generator_any_size_with_and_without_the_256_pixel_tile(lib, pkg):
    """migan-512 at 384 x 640 (forward_any_size), twelve images = two sub-batches of six: the 96 x 160 and 48 x 80 layers have enough 16 x 16
    tiles for sepconv_wide2_kernel; the images must be the bits a batch of two gives (which runs the 128-pixel tiles) and, with the kernel
    switched off, the same again"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("gpu tests need an MI355X (torch.cuda.is_available() is False)")
    dev = torch.device("cuda", 0)
    res, seed = 512, 93
    sd = pkg.synth.make_state_dict(res, seed=seed, regime="export")
    m = pkg.Generator(resolution=res)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to(dev).eval()
    x2 = torch.from_numpy((pkg.synth.normal((2, 4, 384, 640), seed, "xhw") * 0.7).astype(np.float32)).to(dev)
    with torch.no_grad():                                 # (default threshold: 60 tiles x 6 images >= 256 > 60 x 2)
        y2 = m.forward_any_size(x2)
        y12 = m.forward_any_size(x2.repeat(6, 1, 1, 1))
        lib.set_tuning("w2", 0)
        y12_off = m.forward_any_size(x2.repeat(6, 1, 1, 1))
    assert bool(torch.isfinite(y12).all())
    assert torch.equal(y12, y2.repeat(6, 1, 1, 1))
    assert torch.equal(y12, y12_off)
