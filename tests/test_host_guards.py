"""Host-side guards of the C ABI, through the CPU emulator build of the same host code: the 32-bit per-image
offsets of the kernels bound the arbitrary-size forward, a plan that fails to build is never cached, the sub-batch hand-over
event is always recorded, the 'default' GEMM name resolves per storage format."""
import importlib

import numpy as np
import pytest

from tests.emu_util import aligned, emu_lib


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("mi-gan_amd")


def _bind(pkg, lib, res, seed, dtype="f32"):
    h = pkg.hipbind.MiganHandle(lib, res, dtype=dtype)
    sd = pkg.synth.make_state_dict(res, seed=seed, regime="export")
    keep = {k: aligned(v.reshape(1) if v.ndim == 0 else v) for k, v in sd.items()}
    for name, shape, _ in h.weights():
        h.set_weight(name, keep[name].ctypes.data, shape)
    h.commit()
    return h, sd, keep


def test_oversize_forward_hw_is_refused_by_the_offset_width_not_a_pixel_cap(pkg, lib):
    """channels_at(64) = 512: 4096 x 4096 pixels would be 2^33 elements per tensor; the kernels' 32-bit byte offsets end at 2^30
    fp32 elements.  Generator(512) (64 channels at full size) fits a larger image than Generator(64)."""
    h64 = pkg.hipbind.MiganHandle(lib, 64)
    with pytest.raises(ValueError, match="too large"):
        h64.workspace_bytes_hw(1, 4096, 4096)
    with pytest.raises(ValueError, match="too large"):
        h64.workspace_bytes_hw(1, 2048, 1024)                      # 2^21 pixels x 512 channels = 2^30 elements: one past the end
    assert h64.workspace_bytes_hw(1, 1024, 1008) > 0               # just below
    h512 = pkg.hipbind.MiganHandle(lib, 512)
    assert h512.workspace_bytes_hw(1, 2048, 4096) > 0              # 2^23 pixels x 64 channels = 2^29 elements
    with pytest.raises(ValueError, match="too large"):
        h512.workspace_bytes_hw(1, 4096, 4096)


def test_a_failed_plan_is_not_cached(pkg, lib):
    """sizes that are refused leave nothing behind: asking again fails the same way, and a valid size still plans and runs"""
    res = 8
    h, sd, keep = _bind(pkg, lib, res, seed=4)
    for _ in range(2):
        with pytest.raises(ValueError):
            h.workspace_bytes_hw(1, 7, 8)                          # not a multiple of resolution / 4
    for k in range(20):                                            # more distinct sizes than the plan cache keeps
        assert h.workspace_bytes_hw(1, 8 + 2 * (k % 5), 8 + 2 * (k // 5)) > 0
    x = pkg.synth.make_input(1, res, seed=9)
    xa, y = aligned(x), aligned(np.full((1, 3, res, res), np.nan, np.float32))
    need = h.workspace_bytes_hw(1, res, res)
    ws = np.zeros(need // 4 + 64, np.float32)
    h.forward_hw(xa.ctypes.data, y.ctypes.data, 1, res, res, ws.ctypes.data, need)
    assert np.isfinite(y).all()


@pytest.mark.parametrize("pct", [0, 100])
def test_sub_batch_stagger_is_clamped_to_an_existing_launch(pkg, lib, pct):
    """stagger_pct = 100 used to index one past the last launch: the event the next sub-batch waits on was never recorded"""
    res = 8
    lib.set_tuning("stagger_pct", pct)
    try:
        h, sd, keep = _bind(pkg, lib, res, seed=6)
        x = pkg.synth.make_input(16, res, seed=6)
        outs = []
        for streams in (1, 2):
            h.set_streams(streams)
            xa, y = aligned(x), aligned(np.full((16, 3, res, res), np.nan, np.float32))
            need = h.workspace_bytes(16)
            ws = np.zeros(need // 4 + 64, np.float32)
            h.forward(xa.ctypes.data, y.ctypes.data, 16, ws.ctypes.data, need)
            outs.append(y.copy())
        np.testing.assert_array_equal(outs[0], outs[1])
    finally:
        lib.set_tuning("stagger_pct", 22)


def test_default_gemm_name_resolves_per_storage_format(pkg, lib):
    h = pkg.hipbind.MiganHandle(lib, 8)
    h.set_gemm("f32")
    h.set_gemm("default")
    assert h.gemm() == lib.gemm_variant()
    hb = pkg.hipbind.MiganHandle(lib, 8, dtype="bf16")
    hb.set_gemm("f16x2")
    hb.set_gemm("default")
    assert hb.gemm() == "f16"
