"""The software-pipelined persistent SeparableConv2d kernels (mi-gan_amd/csrc/migan_pipe.hpp) on a real MI355X, through the C ABI entry
migan_sepconv_forward, against the numpy oracle: every form (plain, + fused ToRGB, + fused FromRGB, FIR-up), grids where workgroups walk
one, a few and many tiles, run-to-run determinism (the DMA ring and the deferred epilogue must not race)."""
import numpy as np
import pytest
import torch

from tests.sepconv_case import CudaMem, run_sepconv_case

pytestmark = pytest.mark.gpu

PIPE = "migan::sepconv_pipe_kernel<"


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("gpu tests need an MI355X (torch.cuda.is_available() is False)")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def lib(pkg):
    return pkg.load_library()


# both workgroup shapes: 4 or 8 waves in the depthwise group (12- / 16-wave workgroups)
@pytest.fixture(autouse=True, params=[4, 8])
def knobs(request, lib):
    lib.set_tuning("pipe_na", request.param)
    lib.set_tuning("pipe_na8", 0)      # (this wave count for every form)
    yield
    lib.set_tuning("pipe_min_tiles", 256)
    lib.set_tuning("pipe_grid", 256)
    lib.set_tuning("pipe", 15)
    lib.set_tuning("pipe_na", 4)
    lib.set_tuning("pipe_na8", 9)


# (h, w, batch, persistent workgroups): 0 = the default grid (one per CU)
GRIDS = [(64, 64, 8, 0), (16, 32, 5, 8), (128, 128, 3, 0), (64, 128, 2, 64)]


def _grid(lib, grid):
    lib.set_tuning("pipe_min_tiles", 1)
    if grid:
        lib.set_tuning("pipe_grid", grid)


@pytest.mark.parametrize("h,w,batch,grid", GRIDS)
@pytest.mark.parametrize("noise", [False, True])
def test_plain(lib, pkg, dev, h, w, batch, grid, noise):
    _grid(lib, grid)
    run_sepconv_case(lib, pkg, CudaMem(dev), cin=64, cout=64, h=h, w=w, batch=batch, noise=noise, seed=3)
    assert lib.last_kernel().startswith(PIPE + "0, 64, 64, false, false"), lib.last_kernel()


def test_plain_layer_with_a_skip_tensor_keeps_the_one_tile_kernel(lib, pkg, dev):
    """the pipelined plain epilogue has no skip add: default tuning, enough tiles for the pipelined form"""
    run_sepconv_case(lib, pkg, CudaMem(dev), cin=64, cout=64, h=64, w=64, batch=8, noise=True, skip=True, seed=3)
    assert lib.last_kernel().startswith("migan::sepconv_kernel<"), lib.last_kernel()


@pytest.mark.parametrize("h,w,batch,grid", GRIDS)
@pytest.mark.parametrize("prev", [False, True])
def test_plain_with_fused_torgb(lib, pkg, dev, h, w, batch, grid, prev):
    _grid(lib, grid)
    run_sepconv_case(lib, pkg, CudaMem(dev), cin=64, cout=64, h=h, w=w, batch=batch, noise=True, torgb=True, with_prev=prev, seed=5)
    assert lib.last_kernel().startswith(PIPE + "0, 64, 64, false, true"), lib.last_kernel()


@pytest.mark.parametrize("h,w,batch,grid", GRIDS)
def test_plain_with_fused_fromrgb(lib, pkg, dev, h, w, batch, grid):
    _grid(lib, grid)
    run_sepconv_case(lib, pkg, CudaMem(dev), cin=64, cout=64, h=h, w=w, batch=batch, fromrgb=True, seed=7)
    assert lib.last_kernel().startswith(PIPE + "0, 64, 64, true, false"), lib.last_kernel()


@pytest.mark.parametrize("h,w,batch,grid", [(32, 32, 8, 0), (12, 20, 3, 8), (64, 64, 3, 0), (30, 70, 2, 64)])
@pytest.mark.parametrize("noise,skip", [(True, True), (False, False)])
def test_fir_up(lib, pkg, dev, h, w, batch, grid, noise, skip):
    _grid(lib, grid)
    run_sepconv_case(lib, pkg, CudaMem(dev), cin=128, cout=64, h=h, w=w, batch=batch, up=2, noise=noise, skip=skip, seed=9)
    assert lib.last_kernel().startswith(PIPE + "2, 64, 128, false, false"), lib.last_kernel()


def test_pipelined_forward_is_deterministic_and_matches_the_one_tile_kernels(pkg, dev):
    """whole generator at 512: twenty forwards bit-identical, and within fp32 rounding of the plan without the pipelined kernels"""
    lib = pkg.load_library()
    sd = pkg.synth.make_state_dict(512, seed=11)
    x = torch.from_numpy(pkg.synth.make_input(4, 512, seed=11)).to(dev)

    def run():
        m = pkg.Generator(resolution=512)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        m = m.to(dev).eval()
        with torch.no_grad():
            ys = [m(x).clone() for _ in range(20)]
        torch.cuda.synchronize()
        names = [l["kernel"] for l in m._handle.launches()]
        return ys, names

    ys, names = run()
    assert any(n.startswith(PIPE) for n in names), names
    for y in ys[1:]:
        assert torch.equal(y, ys[0])
    lib.set_tuning("pipe", 0)
    try:
        y0, names0 = run()
    finally:
        lib.set_tuning("pipe", 15)
    assert not any(n.startswith(PIPE) for n in names0)
    scale = float(ys[0].abs().max())
    assert float((ys[0] - y0[0]).abs().max()) <= 2e-5 * max(1.0, scale)

DOWN = "migan::sepconv_pipedown_kernel<"


@pytest.mark.parametrize("h,w,batch,grid", [(128, 128, 8, 0), (16, 32, 3, 8), (256, 256, 2, 0), (64, 96, 3, 64)])
@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 256)])
@pytest.mark.parametrize("dna,nb", [(4, 8), (8, 8), (12, 4)])
def test_fused_down(lib, pkg, dev, h, w, batch, grid, cin, cout, dna, nb):
    """down=2 as one launch: depthwise + FIR-down feed the 1x1 through LDS; 4 / 8 / 12 depthwise + FIR waves beside 8 / 8 / 4 GEMM waves"""
    _grid(lib, grid)
    lib.set_tuning("pipe", 15)
    lib.set_tuning("pipe_dna", dna)
    try:
        run_sepconv_case(lib, pkg, CudaMem(dev), cin=cin, cout=cout, h=h, w=w, batch=batch, down=2, seed=11)
        if (cin, dna) == (128, 8):
            dna = 4                                         # (no 8 + 8 instantiation for 128 -> 256: two registers short)
        assert lib.last_kernel() == DOWN + f"{cout}, {cin}, 2, {dna}, {nb}>", lib.last_kernel()
    finally:
        lib.set_tuning("pipe_dna", 12)

@pytest.mark.parametrize("h,w,batch,grid", [(64, 64, 8, 0), (16, 32, 5, 8), (128, 128, 2, 0)])
@pytest.mark.parametrize("torgb", [False, True])
def test_plain_128_to_128(lib, pkg, dev, h, w, batch, grid, torgb):
    """the 256 x 256 layers of migan-512: 128-column tiles (two blocks per B wave), weight planes streamed through the two-slot ring"""
    _grid(lib, grid)
    run_sepconv_case(lib, pkg, CudaMem(dev), cin=128, cout=128, h=h, w=w, batch=batch, noise=True, torgb=torgb, with_prev=torgb, seed=19)
    assert lib.last_kernel().startswith(PIPE + "0, 128, 128, false, " + ("true" if torgb else "false")), lib.last_kernel()
