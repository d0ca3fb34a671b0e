"""The whole launch plan + C ABI (create / set_weight / commit / workspace / forward / debug taps /
timed forward) executed by the CPU fiber emulator on the product's kernel source, against the oracle
and the reference-generated golden vectors."""
import os

import numpy as np
import pytest

from oracle import migan_oracle as orc
from tests.emu_util import aligned, emu_lib


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _bind(pkg, lib, res, seed, regime="export", debug=False):
    h = pkg.hipbind.MiganHandle(lib, res)
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
    ws = np.zeros(h.workspace_bytes(b) // 4 + 64, np.float32)
    h.forward(xa.ctypes.data, y.ctypes.data, b, ws.ctypes.data, h.workspace_bytes(b))
    return y, ws


@pytest.mark.parametrize("name", ["r8_export", "r16_export", "r16_clamp", "r32_init"])
def test_generator_vs_reference_goldens(pkg, lib, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"generator_{name}.npz"))
    r, n, seed = int(g["resolution"]), int(g["batch"]), int(g["seed"])
    h, sd, keep = _bind(pkg, lib, r, seed, regime=str(g["regime"]))
    x = pkg.synth.make_input(n, r, seed=seed, kind=str(g["kind"])) * np.float32(float(g["scale"]))
    y, _ = _forward(h, x)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=3e-5 * max(1.0, float(g["y_absmax"])))


def test_every_layer_vs_oracle_taps(pkg, lib):
    res, batch, seed = 16, 3, 5
    h, sd, keep = _bind(pkg, lib, res, seed, debug=True)
    x = pkg.synth.make_input(batch, res, seed=seed)
    taps = {}
    want = orc.generator(x, sd, res, taps=taps)
    y, ws = _forward(h, x)
    raw = ws.view(np.uint8)
    checked = 0
    for name, ref in taps.items():
        if name.endswith(".conv1") and (name + ".skip") in taps:
            continue
        key = name[:-5] if name.endswith(".skip") else name
        if key == f"synthesis.b{res}.img":
            continue                         # the last running image IS the network output y
        off, shape = h.debug_tensor(batch, key)
        t = raw[off:off + 4 * int(np.prod(shape))].view(np.float32).reshape(shape)
        got = t if key.endswith(".img") else np.transpose(t, (0, 3, 1, 2))
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5 * max(1.0, float(np.abs(ref).max())), err_msg=name)
        checked += 1
    assert checked == 2 * 6 + 2      # 6 blocks x 2 convs + intermediate images b4, b8 (b16 image = y)
    np.testing.assert_allclose(y, want, rtol=0, atol=1e-4)


def test_batch_independence_and_determinism(pkg, lib):
    """Any grouping of two or more images into tiles gives the same per-image result (bit exact).  A single-image forward also runs
    the K-split tiles (round 3: four partial sums per output element), so it equals the same image in a batch to fp32 rounding."""
    res, seed = 8, 6
    h, sd, keep = _bind(pkg, lib, res, seed)
    x = pkg.synth.make_input(5, res, seed=seed)
    y5, _ = _forward(h, x)
    y5b, _ = _forward(h, x)
    y2, _ = _forward(h, x[3:5])
    y1, _ = _forward(h, x[3:4])
    y1b, _ = _forward(h, x[3:4])
    np.testing.assert_array_equal(y5, y5b)
    np.testing.assert_array_equal(y5[3:5], y2)
    np.testing.assert_array_equal(y1, y1b)
    np.testing.assert_allclose(y1, y5[3:4], rtol=0, atol=2e-5 * max(1.0, float(np.abs(y5).max())))


def test_launch_plan_matches_the_survey_accounting(pkg, lib):
    """Per-image algorithmic work the roofline is computed from."""
    h512 = pkg.hipbind.MiganHandle(lib, 512)
    L = h512.launches()
    hb = pkg.hipbind.MiganHandle(lib, 512, dtype="bf16")               # 16-bit activation storage: SURVEY's bf16 byte model
    assert 482.9 <= sum(l["bytes"] for l in hb.launches()) / 1e6 < 482.9 + 6.0      # (network input and RGB planes stay fp32: +4.7 MB)
    seps = [l for l in L if "sepconv_kernel" in l["kernel"] or "sepconv_wide_kernel" in l["kernel"]]
    assert len(seps) == 32                                        # 32 SeparableConv2d @512
    assert len([l for l in L if "dwfir_kernel" in l["kernel"]]) == 7   # one per down=2 layer
    assert abs(sum(l["mfma_flops"] for l in L) / 1e9 - 26.49) < 0.02   # 1x1 convs: 26.49 GFLOP
    assert abs(sum(l["flops"] for l in L) / 1e9 - 29.35) < 0.05        # all stages: 29.35 GFLOP
    assert abs(sum(l["bytes"] for l in L) / 1e6 - 965.7) < 1.0         # 965.7 MB fp32
    h256 = pkg.hipbind.MiganHandle(lib, 256)
    L = h256.launches()
    assert abs(sum(l["mfma_flops"] for l in L) / 1e9 - 20.05) < 0.02
    assert abs(sum(l["bytes"] for l in L) / 1e6 - 455.3) < 1.0
    names = [l["layer"] for l in L]
    assert names[0] == "encoder.b256.conv1" and names[-1] == "synthesis.b256.conv2"


def test_timed_forward_and_workspace_contract(pkg, lib):
    res, seed = 8, 7
    h, sd, keep = _bind(pkg, lib, res, seed)
    x = aligned(pkg.synth.make_input(2, res, seed=seed))
    y = aligned(np.zeros((2, 3, res, res), np.float32))
    need = h.workspace_bytes(2)
    assert h.workspace_bytes(4) > need
    ws = np.zeros(need // 4 + 64, np.float32)
    ms = h.forward_timed(x.ctypes.data, y.ctypes.data, 2, ws.ctypes.data, need)
    assert len(ms) == len(h.launches()) and all(m >= 0 for m in ms)
    with pytest.raises(ValueError):
        h.forward(x.ctypes.data, y.ctypes.data, 2, ws.ctypes.data, need - 256)     # workspace too small


def test_state_errors(pkg, lib):
    h = pkg.hipbind.MiganHandle(lib, 8)
    sd = pkg.synth.make_state_dict(8, seed=1)
    keep = {k: aligned(v.reshape(1) if v.ndim == 0 else v) for k, v in sd.items()}
    x = aligned(np.zeros((1, 4, 8, 8), np.float32))
    ws = np.zeros(h.workspace_bytes(1) // 4 + 64, np.float32)
    with pytest.raises(pkg.MiganError, match="missing key"):
        h.commit()                                                                  # nothing bound yet
    with pytest.raises(ValueError, match="unexpected key"):
        h.set_weight("encoder.b8.nope", keep["encoder.b8.conv1.conv1.bias"].ctypes.data, (512,))
    with pytest.raises(ValueError, match="size mismatch"):
        h.set_weight("encoder.b8.conv1.conv1.bias", keep["encoder.b8.conv1.conv1.bias"].ctypes.data, (256,))
    for name, shape, _ in h.weights():
        h.set_weight(name, keep[name].ctypes.data, shape)
    with pytest.raises(pkg.MiganError, match="before migan_commit"):
        h.forward(x.ctypes.data, x.ctypes.data, 1, ws.ctypes.data, ws.nbytes)
    keep["synthesis.b8.upsample.filter.weight"][...] *= 0.5                         # not setup_filter([1,3,3,1])
    with pytest.raises(NotImplementedError, match="setup_filter"):
        h.commit()
    for r in (12, 4, 8192):                                                         # (not a power of two, below 8, above 4096)
        with pytest.raises(ValueError):
            pkg.hipbind.MiganHandle(lib, r)
    with pytest.raises(NotImplementedError):                                        # above 512: fp32 activation storage only
        pkg.hipbind.MiganHandle(lib, 1024, dtype="bf16")
    h1024 = pkg.hipbind.MiganHandle(lib, 1024)                                      # round 6: 1024 ... 4096 plan (reference :215-223)
    assert sum(l["kernel"].startswith("migan::narrow_sepconv_kernel<") for l in h1024.launches()) == 3


def test_forward_argument_errors(pkg, lib):
    """Empty batch, undersized workspace and null pointers are refused with MIGAN_EINVAL, nothing is launched."""
    h, sd, keep = _bind(pkg, lib, 8, 2)
    x = aligned(pkg.synth.make_input(2, 8, seed=2))
    y = aligned(np.full((2, 3, 8, 8), np.nan, np.float32))
    need = h.workspace_bytes(2)
    ws = np.zeros(need // 4 + 64, np.float32)
    with pytest.raises(ValueError):
        h.workspace_bytes(0)
    with pytest.raises(ValueError, match="batch"):
        h.forward(x.ctypes.data, y.ctypes.data, 0, ws.ctypes.data, need)
    with pytest.raises(ValueError, match="workspace"):
        h.forward(x.ctypes.data, y.ctypes.data, 2, ws.ctypes.data, need - 16)
    with pytest.raises(ValueError):
        h.forward(None, y.ctypes.data, 2, ws.ctypes.data, need)
    with pytest.raises(ValueError):
        h.forward(x.ctypes.data, y.ctypes.data, 2, None, need)
    assert np.isnan(y).all()                                     # none of the refused calls wrote anything
    h.forward(x.ctypes.data, y.ctypes.data, 2, ws.ctypes.data, need)
    np.testing.assert_allclose(y, orc.generator(x, sd, 8), rtol=0, atol=2e-5)
