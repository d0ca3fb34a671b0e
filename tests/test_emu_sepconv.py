"""Operator-level parity of the product kernel source (executed by the CPU fiber emulator,
tests/emu) against the numpy oracle, through the C ABI entry migan_sepconv_forward.
Covers every kernel mode, the small-resolution multi-image tiles, ragged batches, ragged UP
tiles, noise, skip, fused fromrgb and fused ToRGB."""
import numpy as np
import pytest

from oracle import migan_oracle as orc
from tests.emu_util import aligned, emu_lib, nchw, nhwc, ptr


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


# Every test of this file runs under each of these (in-process: the GEMM variant is a field of the operator descriptor, the
# persistent-workgroup thresholds are run-time tuning knobs): the default plan, and the three GEMM variants with persistent
# workgroups forced on the small emulator cases (workgroups walk several tiles and prefetch the next tile's first K chunk
# during the epilogue).
_GEMM = {"code": -1}


@pytest.fixture(autouse=True, params=["default", "f16x2+persist8", "bf16x3+persist16", "f32+persist8"])
def variant(request, lib):
    name = request.param
    if name == "default":
        _GEMM["code"] = -1
        yield name
        return
    gemm, persist = name.split("+persist")
    _GEMM["code"] = {"f32": 0, "bf16x3": 1, "f16x2": 2}[gemm]
    lib.set_tuning("persist_min", 2)
    lib.set_tuning("persist_grid", int(persist))
    yield name
    lib.set_tuning("persist_min", 8192)
    lib.set_tuning("persist_grid", 512)
    _GEMM["code"] = -1


def _weights(pkg, cin, cout, seed, res_out, noise):
    s = pkg.synth
    sd = {
        "m.conv1.weight": (s.normal((cin, 1, 3, 3), seed, "w1") * 0.4).astype(np.float32),
        "m.conv1.bias": (s.normal((cin,), seed, "b1") * 0.5).astype(np.float32),
        "m.conv2.weight": (s.normal((cout, cin, 1, 1), seed, "w2") / np.sqrt(cin)).astype(np.float32),
    }
    if noise:
        sd["m.noise_const"] = s.normal((res_out, res_out), seed, "nc").astype(np.float32)
        sd["m.noise_strength"] = np.asarray(0.37, dtype=np.float32)
    return sd


def _run(lib, pkg, *, cin, cout, res_in, batch, down=1, up=1, noise=False, skip=False, seed=1, wscale=1.0):
    res_out = res_in // 2 if down == 2 else (res_in * 2 if up == 2 else res_in)
    sd = _weights(pkg, cin, cout, seed, res_out, noise)
    sd["m.conv2.weight"] = (sd["m.conv2.weight"] * np.float32(wscale)).astype(np.float32)
    osd = dict(sd)
    if down == 2:
        osd["m.downsample.filter.weight"] = np.broadcast_to(orc.fir_taps(1.0), (cin, 1, 4, 4)).astype(np.float32)
    if up == 2:
        osd["m.upsample.filter.weight"] = np.broadcast_to(orc.fir_taps(4.0), (cout, 1, 4, 4)).astype(np.float32)
    x = (pkg.synth.normal((batch, cin, res_in, res_in), seed, "x") * 1.5).astype(np.float32)
    want = orc.separable_conv(x.copy(), osd, "m")
    sk = None
    if skip:
        sk = pkg.synth.normal((batch, cout, res_out, res_out), seed, "skip").astype(np.float32)
        want = want + sk
    xh = aligned(nhwc(x))
    y = aligned(np.full((batch, res_out, res_out, cout), np.nan, dtype=np.float32))
    skh = aligned(nhwc(sk)) if skip else None
    w1, b1, w2 = aligned(sd["m.conv1.weight"]), aligned(sd["m.conv1.bias"]), aligned(sd["m.conv2.weight"])
    nc = aligned(sd["m.noise_const"]) if noise else None
    ns = aligned(sd["m.noise_strength"].reshape(1)) if noise else None
    scratch = aligned(np.full((batch, res_out, res_out, cin), np.nan, dtype=np.float32)) if down == 2 else None
    wsp = aligned(np.full((3 * cin * cout + 1) // 2 + 8, np.nan, dtype=np.float32))      # bf16 weight planes (bf16x3 GEMM)
    lib.sepconv_forward(x=ptr(xh), y=ptr(y), skip=ptr(skh), conv1_weight=ptr(w1), conv1_bias=ptr(b1),
                        conv2_weight=ptr(w2), noise_const=ptr(nc), noise_strength=ptr(ns),
                        batch=batch, cin=cin, cout=cout, res_in=res_in, down=down, up=up,
                        scratch=ptr(scratch), scratch_bytes=0 if scratch is None else scratch.nbytes,
                        wsplit=ptr(wsp), wsplit_bytes=wsp.nbytes, gemm=_GEMM["code"])
    got = nchw(y)
    assert np.isfinite(got).all(), "kernel left NaNs (unwritten output or read of unwritten LDS)"
    tol = 2e-5 * max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=0, atol=tol)


@pytest.mark.parametrize("cin,cout,res,batch,noise,skip", [
    (64, 64, 16, 1, False, False),     # NT=64, one n-chunk, 2 tiles
    (64, 64, 32, 3, True, True),       # NT=64, 24 tiles: under the persistent re-run 3 tiles per workgroup, resident weight tiles
    (32, 128, 16, 2, True, True),      # NT=128, single K chunk
    (64, 256, 32, 1, True, False),     # wide kernel (8 waves, 128 x 256 tile), 2 K chunks, 8 tiles
    (96, 512, 16, 3, True, True),      # wide kernel, two n-chunks, 3 K chunks, skip, ragged XCD split
    (32, 256, 16, 2, False, True),     # wide kernel, a single K chunk
    (160, 256, 16, 1, True, False),    # wide kernel, 5 K chunks (odd count)
    (64, 384, 16, 1, False, False),    # 3 n-chunks of 128 (Cout not a multiple of 256)
    (64, 64, 8, 3, False, True),       # 2 images per tile, ragged batch
    (64, 128, 4, 3, True, False),      # 8 images per tile, ragged batch
])
def test_plain(lib, pkg, cin, cout, res, batch, noise, skip):
    _run(lib, pkg, cin=cin, cout=cout, res_in=res, batch=batch, noise=noise, skip=skip)


@pytest.mark.parametrize("cin,cout,res_in,batch", [
    (32, 64, 32, 1),      # out 16: 4x16 tiles
    (64, 128, 64, 1),     # out 32: 16 tiles, NT=128, 4 K chunks
    (32, 64, 16, 2),      # out 8: 8x8 tile
    (48 + 16, 128, 8, 5), # out 4: 4 images per tile, ragged batch
])
def test_down(lib, pkg, cin, cout, res_in, batch):
    _run(lib, pkg, cin=cin, cout=cout, res_in=res_in, batch=batch, down=2)


@pytest.mark.parametrize("cin,cout,res_in,batch,noise,skip", [
    (64, 64, 16, 1, True, True),      # out 32: 3x2 ragged tiles
    (128, 64, 16, 2, True, True),     # 4 K chunks, 12 tiles: persistent re-run keeps all four weight tiles resident
    (32, 128, 32, 1, False, False),   # out 64: 6x3 tiles
    (64, 64, 8, 2, True, False),      # out 16: 2x1 tiles
    (32, 128, 4, 3, True, True),      # out 8: 2 images per tile, ragged batch
])
def test_up(lib, pkg, cin, cout, res_in, batch, noise, skip):
    _run(lib, pkg, cin=cin, cout=cout, res_in=res_in, batch=batch, up=2, noise=noise, skip=skip)


@pytest.mark.parametrize("res,batch", [(16, 2), (32, 3)])     # (32, 3): 24 tiles, several per workgroup in the persistent re-run
def test_fromrgb_fused(lib, pkg, res, batch):
    cin = cout = 64
    sd = _weights(pkg, cin, cout, 7, res, False)
    fw = (pkg.synth.normal((cin, 4, 1, 1), 7, "fw") * 0.7).astype(np.float32)
    fb = (pkg.synth.normal((cin,), 7, "fb") * 0.3).astype(np.float32)
    img = pkg.synth.make_input(batch, res, seed=7)
    h = orc.lrelu_agc(orc.pointwise(img, fw, fb))                     # reference :194-195
    want = orc.separable_conv(h, sd, "m")
    x = aligned(img)                                                   # NCHW network input
    y = aligned(np.full((batch, res, res, cout), np.nan, dtype=np.float32))
    arrs = [aligned(a) for a in (sd["m.conv1.weight"], sd["m.conv1.bias"], sd["m.conv2.weight"], fw, fb)]
    wsp = aligned(np.full((3 * cin * cout + 1) // 2 + 8, np.nan, dtype=np.float32))
    lib.sepconv_forward(x=ptr(x), y=ptr(y), conv1_weight=ptr(arrs[0]), conv1_bias=ptr(arrs[1]), conv2_weight=ptr(arrs[2]),
                        fromrgb_weight=ptr(arrs[3]), fromrgb_bias=ptr(arrs[4]), wsplit=ptr(wsp), wsplit_bytes=wsp.nbytes,
                        batch=batch, cin=cin, cout=cout, res_in=res, gemm=_GEMM["code"])
    np.testing.assert_allclose(nchw(y), want, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("with_prev,cout", [(False, 64), (True, 64), (True, 128), (True, 256), (True, 512), (False, 384)])
def test_torgb_fused(lib, pkg, with_prev, cout):
    """ToRGB fused into the epilogue where one workgroup owns all of cout (64, 128, 256), a second launch (torgb_kernel) on y
    where it does not (512, 384): the operator entry the SynthesisBlock sub-module forward goes through"""
    cin = min(cout, 128)
    res, batch = 16, 2
    sd = _weights(pkg, cin, cout, 9, res, True)
    tw = (pkg.synth.normal((3, cout, 1, 1), 9, "tw") / np.sqrt(cout)).astype(np.float32)
    tb = (pkg.synth.normal((3,), 9, "tb") * 0.3).astype(np.float32)
    x = (pkg.synth.normal((batch, cin, res, res), 9, "x")).astype(np.float32)
    prev = pkg.synth.normal((batch, 3, res // 2, res // 2), 9, "prev").astype(np.float32) if with_prev else None
    feat = orc.separable_conv(x.copy(), sd, "m")
    want_img = orc.pointwise(feat, tw, tb)
    if with_prev:
        want_img = orc.upsample2d(prev) + want_img                    # reference :308-313
    xh = aligned(nhwc(x))
    y = aligned(np.full((batch, res, res, cout), np.nan, dtype=np.float32))
    img_out = aligned(np.full((batch, 3, res, res), np.nan, dtype=np.float32))
    arrs = [aligned(a) for a in (sd["m.conv1.weight"], sd["m.conv1.bias"], sd["m.conv2.weight"], sd["m.noise_const"],
                                 sd["m.noise_strength"].reshape(1), tw, tb)]
    pv = aligned(prev) if with_prev else None
    lib.sepconv_forward(x=ptr(xh), y=ptr(y), conv1_weight=ptr(arrs[0]), conv1_bias=ptr(arrs[1]), conv2_weight=ptr(arrs[2]),
                        noise_const=ptr(arrs[3]), noise_strength=ptr(arrs[4]), torgb_weight=ptr(arrs[5]), torgb_bias=ptr(arrs[6]),
                        img_prev=ptr(pv), img_out=ptr(img_out), batch=batch, cin=cin, cout=cout, res_in=res)
    np.testing.assert_allclose(nchw(y), feat, rtol=0, atol=2e-5 * max(1.0, float(np.abs(feat).max())))
    np.testing.assert_allclose(img_out, want_img, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want_img).max())))


@pytest.mark.parametrize("wscale", [3.0e3, 1.7e-5, 0.0])
@pytest.mark.parametrize("kw", [dict(cin=64, cout=64, res_in=16, batch=1), dict(cin=32, cout=128, res_in=8, batch=2, down=2)])
def test_weight_magnitude_does_not_matter(lib, pkg, kw, wscale):
    """The split GEMM variants rescale every 1x1 weight tensor by a power of two derived from its largest
    magnitude (f16x2) so fp16's narrow exponent range never shows: huge, tiny and all-zero weights keep the
    same relative accuracy (large products hit the +-256 clamp of lrelu_agc, small ones stay far from it)."""
    _run(lib, pkg, wscale=wscale, **kw)


def test_bad_arguments_are_rejected(lib):
    a = np.zeros(64, dtype=np.float32)
    with pytest.raises(ValueError):
        lib.sepconv_forward(x=ptr(a), y=ptr(a), conv1_weight=ptr(a), conv1_bias=ptr(a), conv2_weight=ptr(a),
                            batch=1, cin=50, cout=64, res_in=16)          # cin not a multiple of 4 (of 32 for the tiled kernels)
    with pytest.raises(ValueError):
        lib.sepconv_forward(x=ptr(a), y=ptr(a), conv1_weight=ptr(a), conv1_bias=ptr(a), conv2_weight=ptr(a),
                            batch=1, cin=64, cout=64, res_in=0)           # empty image
    with pytest.raises(ValueError):
        lib.sepconv_forward(x=ptr(a), y=ptr(a), conv1_weight=ptr(a), conv1_bias=ptr(a), conv2_weight=ptr(a),
                            batch=1, cin=64, cout=64, res_in=14, width_in=9, down=2, scratch=ptr(a), scratch_bytes=a.nbytes)   # odd size, down=2
    with pytest.raises(ValueError):
        lib.sepconv_forward(x=ptr(a), y=ptr(a), conv1_weight=ptr(a), conv1_bias=ptr(a), conv2_weight=ptr(a),
                            batch=1, cin=64, cout=64, res_in=16, dtype=1)   # 16-bit storage without the f16x2 weight planes
    with pytest.raises(ValueError, match="scratch"):
        lib.sepconv_forward(x=ptr(a), y=ptr(a), conv1_weight=ptr(a), conv1_bias=ptr(a), conv2_weight=ptr(a),
                            batch=1, cin=64, cout=64, res_in=16, down=2)   # down=2 without scratch
    with pytest.raises(ValueError):
        lib.sepconv_forward(x=None, y=ptr(a), conv1_weight=ptr(a), conv1_bias=ptr(a), conv2_weight=ptr(a),
                            batch=1, cin=64, cout=64, res_in=16)          # null tensor
