"""Round-2 features of the kernel library, executed by the CPU fiber emulator on the product's kernel source through the
C ABI: 16-bit activation storage (BASELINE configs[1]), 16-channel K-chunk tiles, two sub-batches, static weights,
per-handle GEMM variants and persistent workgroups (in-process), the arbitrary-size forward and the
uint8-in / uint8-out forward (N2)."""
import os

import numpy as np
import pytest

from oracle import migan_oracle as orc
from oracle import migan_prepost as pp
from tests.emu_util import aligned, emu_lib, from_storage, storage_close
from tests.sepconv_case import HostMem, run_sepconv_case


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.fixture()
def tuned(lib):
    """set process-wide tuning knobs for one test, restore the defaults afterwards"""
    changed = {}
    defaults = dict(kc16=0, kc16_minw=3, w3=3, wide=3, nt256=1, persist_min=8192, persist_grid=512, streams=2, stagger=-1,
                    small=1, small_max_wgs=512, small_kc=64, small_up32=1, small_dwfir=1, small_ksplit=1)

    def set_(key, value):
        changed[key] = True
        lib.set_tuning(key, value)

    yield set_
    for k in changed:
        lib.set_tuning(k, defaults[k])


def _sepconv(lib, pkg, **kw):
    run_sepconv_case(lib, pkg, HostMem(), **kw)


# ------------------------------------------------------------------------------------------------ 16-bit activation storage
@pytest.mark.parametrize("storage", ["bf16", "f16"])
@pytest.mark.parametrize("case", [
    dict(cin=64, cout=64, h=16, batch=2, noise=True, skip=True),                     # NT 64 main tiles
    dict(cin=32, cout=128, h=16, batch=1, noise=True),                                # NT 128
    dict(cin=64, cout=256, h=16, batch=1, noise=True, skip=True),                     # wide kernel
    dict(cin=64, cout=64, h=8, batch=3, skip=True),                                   # 2 images per tile, ragged batch
    dict(cin=64, cout=128, h=4, batch=3, noise=True),                                 # 8 images per tile
    dict(cin=32, cout=64, h=32, batch=1, down=2),                                     # dwfir (16-bit in) + pointwise GEMM (16-bit out)
    dict(cin=64, cout=128, h=8, batch=5, down=2),                                     # small dwfir tiles
    dict(cin=64, cout=64, h=16, batch=1, up=2, noise=True, skip=True),                # FIR-up, skip in 16-bit
    dict(cin=32, cout=128, h=4, batch=3, up=2, noise=True, skip=True),                # FIR-up on 2-image tiles
    dict(cin=64, cout=64, h=16, batch=2, fromrgb=True),                               # fused FromRGB (fp32 planes in, 16-bit out)
    dict(cin=64, cout=64, h=16, batch=2, noise=True, torgb=True, with_prev=True),     # fused ToRGB on the rounded activations
    dict(cin=256, cout=256, h=16, batch=1, noise=True, torgb=True, with_prev=True),   # wide kernel + ToRGB
])
def test_sepconv_16bit_storage(lib, pkg, storage, case):
    _sepconv(lib, pkg, storage=storage, **case)


@pytest.mark.parametrize("storage", ["bf16", "f16"])
@pytest.mark.parametrize("case", [
    dict(cin=64, cout=64, h=16, batch=2, noise=True, skip=True),
    dict(cin=64, cout=256, h=16, batch=1, noise=True, skip=True),                     # wide kernel
    dict(cin=32, cout=64, h=32, batch=1, down=2),
    dict(cin=64, cout=64, h=16, batch=1, up=2, noise=True, skip=True),
    dict(cin=128, cout=128, h=16, batch=2, noise=True, torgb=True, with_prev=True),
])
def test_sepconv_16bit_storage_exact_gemm(lib, pkg, storage, case):
    """16-bit storage with the f16x2 GEMM variant (operands exact) instead of its default "f16" variant."""
    _sepconv(lib, pkg, storage=storage, gemm=2, **case)


def _bind(pkg, lib, res, seed, dtype="f32", debug=False, regime="export"):
    h = pkg.hipbind.MiganHandle(lib, res, dtype=dtype)
    if debug:
        h.set_debug(True)
    sd = pkg.synth.make_state_dict(res, seed=seed, regime=regime)
    keep = {k: aligned(v.reshape(1) if v.ndim == 0 else v) for k, v in sd.items()}
    for name, shape, _ in h.weights():
        h.set_weight(name, keep[name].ctypes.data, shape)
    h.commit()
    return h, sd, keep


def _forward(h, x):
    b, _, r, _ = x.shape
    xa = aligned(x)
    y = aligned(np.full((b, 3, r, r), np.nan, np.float32))
    need = h.workspace_bytes(b)
    ws = np.zeros(need // 4 + 64, np.float32)
    h.forward(xa.ctypes.data, y.ctypes.data, b, ws.ctypes.data, need)
    return y, ws


@pytest.mark.parametrize("storage", ["bf16", "f16"])
def test_generator_16bit_storage_every_layer(pkg, lib, storage):
    """Whole generator with 16-bit feature maps: every stored tensor against the oracle in the same storage mode (one
    rounding step of slack where the fp32 values differ in the last place before rounding), and how far the mode is
    from the fp32 reference."""
    res, batch, seed = 16, 3, 5
    h, sd, keep = _bind(pkg, lib, res, seed, dtype=storage, debug=True)
    x = pkg.synth.make_input(batch, res, seed=seed)
    taps = {}
    want = orc.generator(x, sd, res, taps=taps, storage=storage)
    y, ws = _forward(h, x)
    raw = ws.view(np.uint8)
    checked = 0
    for name, ref in taps.items():
        if name.endswith(".conv1") and (name + ".skip") in taps:
            continue
        key = name[:-5] if name.endswith(".skip") else name
        if key == f"synthesis.b{res}.img":
            continue
        off, shape = h.debug_tensor(batch, key)
        if key.endswith(".img"):
            got = raw[off:off + 4 * int(np.prod(shape))].view(np.float32).reshape(shape)
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-2 * max(1.0, float(np.abs(ref).max())), err_msg=name)
        else:
            t = from_storage(raw[off:off + 2 * int(np.prod(shape))].view(np.uint16).reshape(shape), storage)
            # the first layers agree except for isolated last-place ties; deeper in, an element that rounded the other way
            # moves everything it feeds, so the two computations decorrelate inside the quantisation noise: every stored
            # tensor stays within one storage step of its largest magnitude (+ one step of the element itself)
            # (f16 storage: the fp16 operand rounding of the default GEMM variant is as large as the storage step itself)
            storage_close(np.transpose(t, (0, 3, 1, 2)), ref, storage, ulps=1, frac=0.7 if checked >= 2 else 0.05, what=name,
                          top_ulps=1.0 if storage == "bf16" else 3.0)
        checked += 1
    assert checked == 2 * 6 + 2
    ref32 = orc.generator(x, sd, res)
    err_mode = float(np.abs(want - ref32).max())          # what 16-bit storage costs (oracle vs oracle)
    err = float(np.abs(y - want).max())                   # kernels vs the oracle of the same mode
    scale = float(np.abs(ref32).max())
    print(f"{storage}: |y|max {scale:.2f}  storage-mode error {err_mode:.3e}  kernels vs mode oracle {err:.3e}")
    # tolerance of the mode = its own quantisation noise, measured oracle (16-bit mode) vs oracle (fp32): the kernels sit
    # inside the same envelope, both against the mode oracle and against the fp32 reference
    assert err <= 2.0 * err_mode
    assert float(np.abs(y - ref32).max()) <= 2.0 * err_mode
    assert err_mode <= (2e-2 if storage == "bf16" else 3e-3) * scale


# ------------------------------------------------------------------------------------------------ 16-channel K chunks
@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("case", [
    dict(cin=64, cout=64, h=16, w=32, batch=2, noise=True, torgb=True, with_prev=True),   # last plain layer + ToRGB, 4 K chunks
    dict(cin=96, cout=64, h=16, batch=1, noise=True, skip=True),                            # 6 K chunks (odd 32-blocks)
    dict(cin=64, cout=64, h=16, batch=2, fromrgb=True),                                     # fused FromRGB
    dict(cin=128, cout=64, h=16, batch=1, up=2, noise=True, skip=True),                     # FIR-up, 8 K chunks
])
def test_sepconv_kc16_tiles(lib, pkg, tuned, storage, case):
    tuned("kc16", 7)
    _sepconv(lib, pkg, storage=storage, gemm=2, **case)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("case", [
    dict(cin=64, cout=64, h=16, w=32, batch=2, noise=True, torgb=True, with_prev=True),
    dict(cin=96, cout=64, h=16, batch=1, noise=True, skip=True),
    dict(cin=64, cout=64, h=16, batch=2, fromrgb=True),
    dict(cin=128, cout=64, h=16, batch=1, up=2, noise=True, skip=True),
])
def test_sepconv_three_workgroup_tiles(lib, pkg, tuned, storage, case):
    """the 64-output-channel main tiles built for 3 workgroups per CU (tuning key w3): single-buffered 1x1 weight tile"""
    tuned("w3", 7)
    _sepconv(lib, pkg, storage=storage, gemm=2, **case)


def test_kc16_persistent_workgroups(lib, pkg, tuned):
    tuned("kc16", 7)
    tuned("persist_min", 2)
    tuned("persist_grid", 8)
    _sepconv(lib, pkg, cin=64, cout=64, h=32, batch=3, noise=True, skip=True, gemm=2)       # 24 tiles on 8 workgroups
    _sepconv(lib, pkg, cin=64, cout=64, h=32, batch=3, fromrgb=True, gemm=2)
    _sepconv(lib, pkg, cin=128, cout=64, h=16, batch=2, up=2, noise=True, skip=True, gemm=2)


# ------------------------------------------------------------------------------------------------ GEMM variants, persistence (in-process)
@pytest.mark.parametrize("gemm", ["f32", "bf16x3", "f16x2"])
def test_generator_per_handle_gemm_variant(pkg, lib, tuned, gemm):
    tuned("persist_min", 2)
    tuned("persist_grid", 8)
    res, seed = 32, 4
    h, sd, keep = _bind(pkg, lib, res, seed)
    h.set_gemm(gemm)
    assert h.gemm() == gemm
    x = pkg.synth.make_input(2, res, seed=seed)
    y, _ = _forward(h, x)
    want = orc.generator(x, sd, res)
    np.testing.assert_allclose(y, want, rtol=0, atol=3e-5 * max(1.0, float(np.abs(want).max())))
    kernels = " ".join(l["kernel"] for l in h.launches())
    assert {"f32": ", 0, false, 0>", "bf16x3": ", 1, false, 0>", "f16x2": ", 2, false, 0>"}[gemm] in kernels
    if gemm != "f16x2":
        assert "wide" not in kernels
    with pytest.raises(ValueError):
        h.set_gemm(7)
    hb = pkg.hipbind.MiganHandle(lib, 8, dtype="bf16")
    assert hb.gemm() == "f16"                                      # default of the 16-bit storage modes
    with pytest.raises(ValueError, match="f16x2"):
        hb.set_gemm("f32")
    hb.set_gemm("f16x2")
    assert hb.gemm() == "f16x2"
    with pytest.raises(ValueError):
        h.set_gemm("f16")                                          # fp32 storage keeps fp32-grade products


# ------------------------------------------------------------------------------------------------ small-launch tiles (round 3)
@pytest.mark.parametrize("knobs,expect", [
    (dict(small=0), ["<0, 128, 128, 32, false, 9, 2, false", "<2, 128, 128, 32, false, 9, 2, false", "<3, 128, 128, 32, false, 4, 2, false"]),
    (dict(small_kc=32, small_up32=0), ["<0, 32, 128, 32, false, 3, 2", "<2, 64, 128, 32, false, 4, 2", "<3, 32, 128, 32, false, 1, 2"]),
    (dict(small_kc=64, small_up32=0), ["<0, 32, 128, 64, false, 5, 2", "<2, 64, 128, 64, false, 7, 2", "<3, 32, 128, 64, false, 2, 2"]),
    (dict(small_kc=32), ["<2, 32, 128, 32, false, 2, 2"]),
    (dict(), ["<0, 32, 128, 64, false, 5, 2", "<2, 32, 128, 64, false, 4, 2", "<3, 32, 128, 64, false, 2, 2"]),
    (dict(small_max_wgs=6), ["<0, 128, 128, 32, false, 9, 2, false", "<0, 32, 128, 64, false, 5, 2"]),   # per launch: only the <= 6-tile launches
    (dict(small_dwfir=0, small_ksplit=0), ["<0, 32, 128, 64, false, 5, 2"]),                             # four channel chunks per dwfir workgroup
], ids=["regular", "kc32", "kc64", "kc32_up32", "default", "threshold", "no_dwfir_split"])
def test_generator_small_launch_tiles(pkg, lib, tuned, golden_dir, knobs, expect):
    """Launches of few workgroups run 32-row (FIR-up: 32- or 64-row) tiles with 32- or 64-channel K chunks; every combination against
    the reference golden of a generator whose layers are all "small" (R = 32, 512 channels everywhere below 64x64), plus a batch
    that is not a multiple of the two-image b4 tile"""
    for k, v in knobs.items():
        tuned(k, v)
    g = np.load(os.path.join(golden_dir, "generator_r32_init.npz"))
    r, n, seed = int(g["resolution"]), int(g["batch"]), int(g["seed"])
    h, sd, keep = _bind(pkg, lib, r, seed, regime=str(g["regime"]))
    x = pkg.synth.make_input(n, r, seed=seed, kind=str(g["kind"])) * np.float32(float(g["scale"]))
    y, _ = _forward(h, x)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=3e-5 * max(1.0, float(g["y_absmax"])))
    kernels = " ".join(l["kernel"] for l in h.launches())
    for e in expect:
        assert e in kernels, (e, kernels)
    x3 = pkg.synth.make_input(3, r, seed=seed + 1)
    y3, _ = _forward(h, x3)
    want = orc.generator(x3, sd, r)
    np.testing.assert_allclose(y3, want, rtol=0, atol=3e-5 * max(1.0, float(np.abs(want).max())))
    assert ", 32, 32, 64, " not in " ".join(l["kernel"] for l in h.launches())       # K-split tiles: single-image forwards only
    # a single image: the 32 x 32 K-split tiles (four waves, four partial sums) where the knobs allow them
    y1, _ = _forward(h, x3[:1])
    np.testing.assert_allclose(y1, want[:1], rtol=0, atol=3e-5 * max(1.0, float(np.abs(want).max())))
    ksplit = ", 32, 32, 64, " in " ".join(l["kernel"] for l in h.launches())
    assert ksplit == (knobs.get("small", 1) == 1 and knobs.get("small_kc", 64) == 64 and knobs.get("small_max_wgs", 512) >= 16
                      and knobs.get("small_ksplit", 1) == 1)
    if not ksplit:
        np.testing.assert_array_equal(y1[0], y3[0])                                   # same summation order: bit-identical


# ------------------------------------------------------------------------------------------------ two sub-batches
def test_two_sub_batches_equal_one_batch(pkg, lib):
    """A forward of >= 16 images is planned as two sub-batches (two streams on the GPU); every image is computed exactly as
    in a single-batch forward, whatever the grouping (bit-identical), including a ragged second half."""
    res, seed = 8, 6
    h, sd, keep = _bind(pkg, lib, res, seed)
    x = pkg.synth.make_input(21, res, seed=seed)
    h.set_streams(1)
    need1 = h.workspace_bytes(21)
    y1, _ = _forward(h, x)
    h.set_streams(2)
    assert h.workspace_bytes(21) >= need1
    y2, _ = _forward(h, x)
    np.testing.assert_array_equal(y1, y2)
    x37 = pkg.synth.make_input(37, res, seed=seed + 1)
    h.set_streams(1)
    y37, _ = _forward(h, x37)
    for n in (3, 4):                                                # 16 + 16 + 5 and 16 + 16 + 5 (whole groups of 8, remainder last)
        h.set_streams(n)
        yn, _ = _forward(h, x37)
        np.testing.assert_array_equal(yn, y37)
    with pytest.raises(ValueError):
        h.set_streams(5)
    h.set_streams(2)
    np.testing.assert_allclose(y2[17:19], orc.generator(x[17:19], sd, res), rtol=0, atol=2e-5)
    # the timed path (one stream, whole-batch launches) fits the same workspace
    xa, ya = aligned(x), aligned(np.full((21, 3, res, res), np.nan, np.float32))
    need = h.workspace_bytes(21)
    ws = np.zeros(need // 4 + 64, np.float32)
    ms = h.forward_timed(xa.ctypes.data, ya.ctypes.data, 21, ws.ctypes.data, need)
    assert len(ms) == len(h.launches())
    np.testing.assert_array_equal(ya, y1)


def test_forward_parts_hands_each_sub_batch_its_own_output(pkg, lib):
    """migan_forward_parts (multi-GPU callers, include/migan_hip.h): the same forward with the sub-batch hand-over made explicit -- sub-batch
    k writes to y_parts[k] (here: separate buffers, as the slices of a collective's receive buffers are) and the split is the one
    migan_forward_split reports; images are the bits migan_forward gives."""
    res, seed = 8, 6
    h, sd, keep = _bind(pkg, lib, res, seed)
    for batch, streams, want_split in ((21, 2, [16, 5]), (37, 3, [16, 16, 5]), (7, 2, [7]), (32, 1, [32])):
        h.set_streams(streams)
        assert h.forward_split(batch) == want_split
        x = pkg.synth.make_input(batch, res, seed=seed + batch)
        y, _ = _forward(h, x)
        xa = aligned(x)
        outs = [aligned(np.full((n, 3, res, res), np.nan, np.float32)) for n in want_split]
        need = h.workspace_bytes(batch)
        ws = np.zeros(need // 4 + 64, np.float32)
        got = h.forward_parts(xa.ctypes.data, [o.ctypes.data for o in outs], batch, ws.ctypes.data, need, 0, [1] * (len(want_split) - 1))
        assert got == want_split
        np.testing.assert_array_equal(np.concatenate(outs, axis=0), y)
    h.set_streams(2)
    xa = aligned(pkg.synth.make_input(21, res, seed=1))
    outs = [aligned(np.zeros((n, 3, res, res), np.float32)) for n in (16, 5)]
    need = h.workspace_bytes(21)
    ws = np.zeros(need // 4 + 64, np.float32)
    with pytest.raises(ValueError):          # a caller stream per sub-batch after the first is required
        h.forward_parts(xa.ctypes.data, [o.ctypes.data for o in outs], 21, ws.ctypes.data, need, 0, [])


# ------------------------------------------------------------------------------------------------ static weights
def test_static_weights_contract(pkg, lib):
    res, seed = 8, 3
    h, sd, keep = _bind(pkg, lib, res, seed)
    x = pkg.synth.make_input(2, res, seed=seed)
    xa = aligned(x)
    need = h.workspace_bytes(2)
    ws = np.zeros(need // 4 + 64, np.float32)
    ws2 = np.zeros(need // 4 + 64, np.float32)

    def run(w=ws):
        y = aligned(np.full((2, 3, res, res), np.nan, np.float32))
        h.forward(xa.ctypes.data, y.ctypes.data, 2, w.ctypes.data, need)
        return y

    y0 = run()
    h.assume_static_weights(True)
    np.testing.assert_array_equal(run(), y0)                      # prepares the planes once more ...
    np.testing.assert_array_equal(run(), y0)                      # ... and reuses them
    name = "synthesis.b8.conv2.conv2.weight"
    keep[name][...] *= np.float32(0.5)                            # in-place write behind the library's back
    np.testing.assert_array_equal(run(), y0)                      # NOT seen: that is the documented contract of the assertion
    y_new_ws = run(ws2)                                           # another workspace has no planes yet: prepared from the new values
    assert np.abs(y_new_ws - y0).max() > 1e-3
    h.assume_static_weights(True)                                 # documented way to invalidate
    y1 = run()
    np.testing.assert_array_equal(y1, y_new_ws)
    sd2 = dict(sd)
    sd2[name] = keep[name]
    np.testing.assert_allclose(y1, orc.generator(x, sd2, res), rtol=0, atol=2e-5 * max(1.0, float(np.abs(y1).max())))
    keep[name][...] *= np.float32(2.0)
    h.set_weight(name, keep[name].ctypes.data, keep[name].shape)   # re-binding invalidates too
    h.commit()
    np.testing.assert_array_equal(run(), y0)
    h.assume_static_weights(False)
    keep[name][...] *= np.float32(0.5)
    np.testing.assert_array_equal(run(), y_new_ws)                # without the assertion in-place updates are always seen


# ------------------------------------------------------------------------------------------------ arbitrary-size forward (N4)
@pytest.mark.parametrize("res,hw,storage", [(16, (12, 20), "f32"), (16, (4, 8), "f32"), (32, (24, 40), "f32"), (16, (20, 12), "bf16"),
                                            (16, (32, 48), "f32")])
def test_forward_hw_vs_oracle(pkg, lib, res, hw, storage):
    """Fully convolutional forward: ragged 8x16 tiles at every level, noise planes tiled / cropped, H != W."""
    hh, ww = hw
    seed, batch = 11, 2
    h, sd, keep = _bind(pkg, lib, res, seed, dtype=storage)
    x = (pkg.synth.normal((batch, 4, hh, ww), seed, "xhw") * 0.7).astype(np.float32)
    want = orc.generator(x, sd, res, storage=storage)
    xa = aligned(x)
    y = aligned(np.full((batch, 3, hh, ww), np.nan, np.float32))
    need = h.workspace_bytes_hw(batch, hh, ww)
    ws = np.zeros(need // 4 + 64, np.float32)
    h.forward_hw(xa.ctypes.data, y.ctypes.data, batch, hh, ww, ws.ctypes.data, need)
    assert np.isfinite(y).all()
    tol = (3e-5 if storage == "f32" else 2e-2) * max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(y, want, rtol=0, atol=tol)
    # the fixed-size forward still runs on the same handle, and H = W = resolution through forward_hw is the same plan
    x0 = pkg.synth.make_input(1, res, seed=seed)
    y0, _ = _forward(h, x0)
    y1 = aligned(np.full((1, 3, res, res), np.nan, np.float32))
    need = h.workspace_bytes_hw(1, res, res)
    ws = np.zeros(need // 4 + 64, np.float32)
    h.forward_hw(aligned(x0).ctypes.data, y1.ctypes.data, 1, res, res, ws.ctypes.data, need)
    np.testing.assert_array_equal(y0, y1)
    for bad in ((hh + 1, ww), (0, ww), (hh, ww + 2)):
        with pytest.raises(ValueError, match="multiples"):
            h.workspace_bytes_hw(batch, *bad)


# ------------------------------------------------------------------------------------------------ uint8 in / uint8 out (N2)
@pytest.mark.parametrize("res", [16, 64])
def test_forward_u8_equals_pack_forward_compose(pkg, lib, res):
    """preprocess() fused into the first kernel and postprocess + compose fused into the last ToRGB epilogue: the same
    bytes as the three-step path, which tests/test_prepost.py pins to the reference's own preprocess()."""
    seed, batch = 13, 2
    h, sd, keep = _bind(pkg, lib, res, seed)
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(batch, res, res, 3), dtype=np.uint8)
    mask = np.where(rng.random((batch, res, res)) < 0.4, 0, 255).astype(np.uint8)
    mask[0, :3, :5] = 128                                            # neither 0 nor 255 = hole (demo.py:60)
    x = aligned(np.zeros((batch, 4, res, res), np.float32))
    lib.pack_input(img.ctypes.data, mask.ctypes.data, x.ctypes.data, batch, res)
    np.testing.assert_array_equal(x, pp.preprocess(img, mask))
    y, _ = _forward(h, x)
    want = np.zeros_like(img)
    lib.compose_output(y.ctypes.data, img.ctypes.data, mask.ctypes.data, want.ctypes.data, batch, res)
    np.testing.assert_array_equal(want, pp.compose(y, img, mask))
    out = np.full_like(img, 77)
    need = h.workspace_bytes(batch)
    ws = np.zeros(need // 4 + 64, np.float32)
    h.forward_u8(img.ctypes.data, mask.ctypes.data, out.ctypes.data, batch, ws.ctypes.data, need)
    np.testing.assert_array_equal(out, want)
    assert (out[mask == 255] == img[mask == 255]).all()


def test_forward_hw_vs_reference_goldens(pkg, lib, golden_dir):
    """the reference module itself run at H x W with dynamic buffers (tests/golden/make_golden_hw.py)"""
    import glob
    import os
    for f in sorted(glob.glob(os.path.join(golden_dir, "generator_hw_r[13]*.npz"))):
        g = np.load(f)
        r, n, seed, hh, ww = int(g["resolution"]), int(g["batch"]), int(g["seed"]), int(g["height"]), int(g["width"])
        h, sd, keep = _bind(pkg, lib, r, seed)
        x = aligned((pkg.synth.normal((n, 4, hh, ww), seed, "xhw") * 0.7).astype(np.float32))
        y = aligned(np.full((n, 3, hh, ww), np.nan, np.float32))
        need = h.workspace_bytes_hw(n, hh, ww)
        ws = np.zeros(need // 4 + 64, np.float32)
        h.forward_hw(x.ctypes.data, y.ctypes.data, n, hh, ww, ws.ctypes.data, need)
        np.testing.assert_allclose(y, g["y"], rtol=0, atol=3e-5 * max(1.0, float(g["y_absmax"])), err_msg=os.path.basename(f))


# ------------------------------------------------------------------------------------------------ wide tile, all MFMAs on waves 4-7 (round 3)
@pytest.mark.parametrize("storage,gemm", [("f32", 2), ("bf16", -1), ("f16", 2)])
@pytest.mark.parametrize("case", [
    dict(cin=64, cout=256, h=16, batch=2, noise=True, skip=True),
    dict(cin=96, cout=512, h=16, w=32, batch=1, noise=True),
    dict(cin=64, cout=256, h=16, batch=1, noise=True, torgb=True, with_prev=True),
])
@pytest.mark.parametrize("wide", [2, 3])
def test_sepconv_wide_tile_with_dedicated_mfma_waves(lib, pkg, storage, gemm, case, wide):
    lib.set_tuning("wide", wide)          # 3: + LDS-DMA staging (fp32 storage; other storage formats keep the register path)
    try:
        _sepconv(lib, pkg, storage=storage, gemm=gemm, **case)
    finally:
        lib.set_tuning("wide", 3)


# ------------------------------------------------------------------------------------------------ FIR-up layers on the 256-column tile (round 4)
WIDE_UP = "migan::sepconv_wide_kernel<false, 0, false, true, true, true>"


@pytest.mark.parametrize("case", [
    dict(cin=64, cout=256, h=16, batch=2, noise=True, skip=True),            # 3 x 2 tiles of 6 x 14 interior pixels, ragged on both edges
    dict(cin=96, cout=512, h=12, w=20, batch=1, noise=True),                 # two column chunks, three K chunks, not a multiple of the tile
    dict(cin=32, cout=256, h=8, w=16, batch=3, skip=True),                   # one K chunk, no noise
    dict(cin=160, cout=256, h=6, w=14, batch=1, noise=True, skip=True),      # exactly one tile: every halo pixel lies outside the image
])
def test_fir_up_on_the_wide_tile(lib, pkg, case):
    """up=2 SeparableConv2d with Cout % 256 == 0 (fp32 storage, f16x2 GEMM): sepconv_wide_kernel<..., UP>; off -> the 128-column kernel"""
    _sepconv(lib, pkg, storage="f32", gemm=2, up=2, **case)
    assert lib.last_kernel() == WIDE_UP, lib.last_kernel()
    lib.set_tuning("wide_up", 0)
    try:
        _sepconv(lib, pkg, storage="f32", gemm=2, up=2, **case)
        assert lib.last_kernel().startswith("migan::sepconv_kernel<2, "), lib.last_kernel()
    finally:
        lib.set_tuning("wide_up", 1)
