"""Round-2 parity tests on a real MI355X, through the C ABI: 16-bit activation storage (BASELINE configs[1]), 16-channel
K-chunk tiles, GEMM variants per handle (in-process), two-stream sub-batches, static weights, the arbitrary-size
forward, the uint8-in / uint8-out forward (N2) and a batch of 32 distinct images."""
import numpy as np
import pytest
import torch

from oracle import migan_oracle as orc
from oracle import migan_prepost as pp
from oracle import migan_torch_cpu as torc
from tests.sepconv_case import CudaMem, run_sepconv_case

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("gpu tests need an MI355X (torch.cuda.is_available() is False)")
    return torch.device("cuda", 0)


@pytest.fixture()
def tuned(pkg):
    lib = pkg.load_library()
    changed = {}
    defaults = dict(kc16=0, kc16_minw=3, w3=3, wide=3, nt256=1, persist_min=8192, persist_grid=512, streams=2, stagger=-1,
                    small=1, small_max_wgs=512, small_kc=64, small_up32=1, small_dwfir=1, small_ksplit=1, pipe=15, wide_up=1)

    def set_(key, value):
        changed[key] = True
        lib.set_tuning(key, value)

    yield set_
    for k in changed:
        lib.set_tuning(k, defaults[k])


def _model(pkg, res, seed, dev, regime="export", **kw):
    sd = pkg.synth.make_state_dict(res, seed=seed, regime=regime)
    m = pkg.Generator(resolution=res, **kw)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


OPERATOR_CASES = [
    dict(cin=64, cout=64, h=16, batch=2, noise=True, skip=True),
    dict(cin=32, cout=128, h=16, batch=1, noise=True),
    dict(cin=64, cout=256, h=32, batch=1, noise=True, skip=True),
    dict(cin=96, cout=512, h=16, batch=3, noise=True, skip=True),
    dict(cin=64, cout=64, h=8, batch=3, skip=True),
    dict(cin=64, cout=128, h=4, batch=3, noise=True),
    dict(cin=32, cout=64, h=32, batch=1, down=2),
    dict(cin=64, cout=128, h=64, batch=1, down=2),
    dict(cin=64, cout=128, h=8, batch=5, down=2),
    dict(cin=64, cout=64, h=16, batch=1, up=2, noise=True, skip=True),
    dict(cin=32, cout=128, h=32, batch=1, up=2),
    dict(cin=32, cout=128, h=4, batch=3, up=2, noise=True, skip=True),
    dict(cin=64, cout=64, h=32, batch=2, fromrgb=True),
    dict(cin=64, cout=64, h=16, batch=2, noise=True, torgb=True, with_prev=True),
    dict(cin=128, cout=128, h=16, batch=2, noise=True, torgb=True, with_prev=True),
    dict(cin=256, cout=256, h=32, batch=1, noise=True, torgb=True, with_prev=True),
]


# ------------------------------------------------------------------------------------------------ GEMM variants, in process
@pytest.mark.parametrize("gemm", [0, 1, 2])
@pytest.mark.parametrize("case", OPERATOR_CASES)
def test_sepconv_every_gemm_variant(pkg, dev, gemm, case):
    """exact fp32 MFMA (0), bf16x3-split (1) and f16x2-split (2) through the descriptor's `gemm` field."""
    run_sepconv_case(pkg.load_library(), pkg, CudaMem(dev), gemm=gemm, **case)


@pytest.mark.parametrize("gemm", ["f32", "bf16x3", "f16x2"])
def test_generator_per_handle_gemm_variant(pkg, dev, gemm):
    for res, batch, seed in ((64, 3, 23), (256, 2, 31)):
        m, sd = _model(pkg, res, seed, dev)
        m.set_gemm(gemm)
        x = pkg.synth.make_input(batch, res, seed=seed)
        with torch.no_grad():
            y = m(torch.from_numpy(x).to(dev)).cpu()
        assert m._handle.gemm() == gemm
        want = torc.generator(x, sd, res)
        assert float((y - want).abs().max()) <= TOL


# ------------------------------------------------------------------------------------------------ 16-bit activation storage
@pytest.mark.parametrize("storage", ["bf16", "f16"])
@pytest.mark.parametrize("case", OPERATOR_CASES)
def test_sepconv_16bit_storage(pkg, dev, storage, case):
    run_sepconv_case(pkg.load_library(), pkg, CudaMem(dev), storage=storage, **case)


@pytest.mark.parametrize("storage", ["bf16", "f16"])
@pytest.mark.parametrize("case", OPERATOR_CASES[::2])
def test_sepconv_16bit_storage_exact_gemm(pkg, dev, storage, case):
    """16-bit storage with the f16x2 GEMM variant (operands exact) instead of its default "f16" variant."""
    run_sepconv_case(pkg.load_library(), pkg, CudaMem(dev), storage=storage, gemm=2, **case)


@pytest.mark.parametrize("res,storage,gemm", [(256, "bf16", "f16"), (512, "bf16", "f16"), (256, "f16", "f16"), (64, "bf16", "f16"),
                                              (256, "bf16", "f16x2")])
def test_generator_16bit_storage_full_size(pkg, dev, res, storage, gemm):
    """BASELINE configs[1] (migan-256, bf16 storage) and the same mode at 512: the tolerance of a storage mode is its own
    quantisation noise, measured oracle(mode) vs oracle(fp32) on the same inputs; the kernels must sit inside that
    envelope both against the mode's oracle and against the fp32 reference (tests/test_emu_round2.py checks every stored
    tensor of a small generator to one storage step)."""
    seed, batch = 33, 2
    m, sd = _model(pkg, res, seed, dev, activation_dtype=storage)
    assert m.activation_dtype == storage
    m.set_gemm(gemm)
    x = pkg.synth.make_input(batch, res, seed=seed)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu()
        y2 = m(torch.from_numpy(x).to(dev)).cpu()
    assert torch.equal(y, y2) and m._handle.gemm() == gemm
    want_mode = torc.generator(x, sd, res, storage=storage, gemm16=(gemm == "f16"))
    want_f32 = torc.generator(x, sd, res)
    err_mode = float((want_mode - want_f32).abs().max())
    err = float((y - want_mode).abs().max())
    err32 = float((y - want_f32).abs().max())
    print(f"migan-{res} {storage} gemm {gemm}: |y|max {float(want_f32.abs().max()):.2f}  mode vs fp32 oracle {err_mode:.3e}  "
          f"kernels vs mode oracle {err:.3e}  kernels vs fp32 oracle {err32:.3e}")
    assert err <= 2.0 * err_mode and err32 <= 2.0 * err_mode
    assert err_mode <= (3e-2 if storage == "bf16" else 4e-3) * float(want_f32.abs().max())
    kernels = " ".join(l["kernel"] for l in m.launch_info())
    assert (", 1>" if storage == "bf16" else ", 2>") in kernels and ", 0>" not in kernels.replace("torgb_kernel<0>", "")


@pytest.mark.parametrize("tag", ["bf16_r64", "bf16_r64_x2", "f16_r64", "bf16_r256", "bf16_r256_x2"])
def test_generator_16bit_storage_against_the_hooked_reference(pkg, dev, golden_dir, tag):
    """BASELINE configs[1] held to REFERENCE-produced numbers: tests/golden/make_golden_bf16.py ran the
    reference module with forward hooks that round every stored feature map to the storage format (and the 1x1 operands to fp16 for the
    "f16" GEMM variant).  The fixture carries that output, the plain fp32 output of the same module and hence the mode's quantisation
    envelope AS THE REFERENCE SHOWS IT.  A rounding step turns a 1-ulp difference in summation order into a full quantisation step, so
    two correct implementations of the mode agree to the noise level, not to the bit (tests/test_oracle_golden.py shows it for the numpy
    oracle; the torch-CPU oracle, same op order as the reference, is bit-equal).  Stated tolerance, absolute, per fixture:
    max|y_hip - y_ref_mode| <= 2 x envelope (0.23 / 0.19 / 0.046 / 0.33 / 0.35 on |y|max 16.4 ... 16.7), rms <= 2 x envelope_rms,
    and against the reference's FP32 output the same bounds (the kernels may not be noisier than the mode itself)."""
    import os
    g = np.load(os.path.join(golden_dir, f"storage_{tag}.npz"))
    res, batch, seed, s = int(g["resolution"]), int(g["batch"]), int(g["seed"]), int(g["stride"])
    storage, gemm = str(g["storage"]), ("f16" if int(g["gemm16"]) else "f16x2")
    m, _ = _model(pkg, res, seed, dev, activation_dtype=storage)
    m.set_gemm(gemm)
    x = pkg.synth.make_input(batch, res, seed=seed)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()[:, :, ::s, ::s].astype(np.float64)
    env, env_rms = float(g["envelope"]), float(g["envelope_rms"])
    d_mode, d_f32 = y - g["y"], y - g["y_f32"]
    print(f"storage_{tag}: reference envelope max {env:.3e} rms {env_rms:.3e};  kernels vs hooked reference max {np.abs(d_mode).max():.3e} "
          f"rms {np.sqrt((d_mode ** 2).mean()):.3e};  vs the fp32 reference max {np.abs(d_f32).max():.3e} rms {np.sqrt((d_f32 ** 2).mean()):.3e}")
    assert np.abs(d_mode).max() <= 2.0 * env and np.sqrt((d_mode ** 2).mean()) <= 2.0 * env_rms
    assert np.abs(d_f32).max() <= 2.0 * env and np.sqrt((d_f32 ** 2).mean()) <= 1.5 * env_rms


# ------------------------------------------------------------------------------------------------ 16-channel K chunks
@pytest.mark.parametrize("minw", [2, 3, 4])
@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("case", [
    dict(cin=64, cout=64, h=32, w=64, batch=2, noise=True, torgb=True, with_prev=True),
    dict(cin=96, cout=64, h=32, batch=1, noise=True, skip=True),
    dict(cin=64, cout=64, h=32, batch=2, fromrgb=True),
    dict(cin=128, cout=64, h=32, batch=1, up=2, noise=True, skip=True),
])
def test_sepconv_kc16_tiles(pkg, dev, tuned, minw, storage, case):
    tuned("kc16", 7)
    tuned("kc16_minw", minw)
    run_sepconv_case(pkg.load_library(), pkg, CudaMem(dev), storage=storage, gemm=2, **case)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("case", [
    dict(cin=64, cout=64, h=32, w=64, batch=2, noise=True, torgb=True, with_prev=True),
    dict(cin=96, cout=64, h=32, batch=1, noise=True, skip=True),
    dict(cin=64, cout=64, h=32, batch=2, fromrgb=True),
    dict(cin=128, cout=64, h=32, batch=1, up=2, noise=True, skip=True),
])
def test_sepconv_three_workgroup_tiles(pkg, dev, tuned, storage, case):
    tuned("w3", 7)
    run_sepconv_case(pkg.load_library(), pkg, CudaMem(dev), storage=storage, gemm=2, **case)


def test_generator_512_with_three_workgroup_tiles(pkg, dev, tuned):
    tuned("w3", 7)
    tuned("pipe", 0)            # (the default plan runs these layers on the pipelined kernels; the one-tile forms stay covered here)
    res, seed, batch = 512, 32, 2
    m, sd = _model(pkg, res, seed, dev)
    x = pkg.synth.make_input(batch, res, seed=seed)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu()
    assert ", 6, 3, true, false, 2, " in " ".join(l["kernel"] for l in m.launch_info())
    assert float((y - torc.generator(x, sd, res)).abs().max()) <= TOL


def test_generator_512_with_kc16_tiles(pkg, dev, tuned):
    tuned("kc16", 7)
    tuned("pipe", 0)
    res, seed, batch = 512, 32, 2
    m, sd = _model(pkg, res, seed, dev)
    x = pkg.synth.make_input(batch, res, seed=seed)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu()
    assert ", 16, " in " ".join(l["kernel"] for l in m.launch_info())
    assert float((y - torc.generator(x, sd, res)).abs().max()) <= TOL


# ------------------------------------------------------------------------------------------------ small-launch tiles, batch-1 latency path
@pytest.mark.parametrize("knobs,expect", [
    (dict(small=0), "<0, 128, 128, 32, false, 9, 2, false"),
    (dict(small_kc=32, small_up32=0), "<2, 64, 128, 32, false, 4, 2"),
    (dict(small_kc=64, small_up32=0), "<2, 64, 128, 64, false, 7, 2"),
    (dict(small_kc=32), "<2, 32, 128, 32, false, 2, 2"),
    (dict(small_ksplit=0), "<0, 32, 128, 64, false, 5, 2"),
    (dict(small_dwfir=0), "<2, 32, 32, 64, false, 4, 2"),
    (dict(), "<0, 32, 32, 64, false, 5, 2"),
], ids=["regular", "kc32", "kc64", "kc32_up32", "no_ksplit", "no_dwfir_split", "default"])
def test_generator_512_batch1_small_launch_tiles(pkg, dev, tuned, knobs, expect):
    """batch 1 at 512x512 (how scripts/demo.py:122-134 calls the model): every layer from 4x4 to 64x64 runs the small-launch tiles;
    parity against the torch-CPU port within the north star's tolerance, for every tile / chunk combination"""
    for k, v in knobs.items():
        tuned(k, v)
    res, seed = 512, 33
    m, sd = _model(pkg, res, seed, dev)
    x = pkg.synth.make_input(1, res, seed=seed, kind="demo")
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu()
    assert expect in " ".join(l["kernel"] for l in m.launch_info())
    assert float((y - torc.generator(x, sd, res)).abs().max()) <= TOL


def test_small_launch_tiles_do_not_change_results_across_batch_sizes(pkg, dev):
    """which tile a layer runs on depends on the launch size; an image must not: image 0 of a batch-1, batch-3 and batch-32 forward"""
    res, seed = 256, 34
    m, sd = _model(pkg, res, seed, dev)
    x = torch.from_numpy(pkg.synth.make_input(32, res, seed=seed, kind="demo")).to(dev)
    with torch.no_grad():
        y32 = m(x).cpu()
        y3 = m(x[:3].contiguous()).cpu()
        y1 = m(x[:1].contiguous()).cpu()
    ref = torc.generator(x[:3].cpu().numpy(), sd, res)
    assert float((y3 - ref).abs().max()) <= TOL
    # batches of two or more: different tiles sum the same K in the same order per output element -> bit-identical
    assert torch.equal(y3, y32[:3])
    # a single-image forward also runs the K-split tiles (four partial sums per output element): the same image to fp32 rounding
    assert float((y1[0] - y3[0]).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert "<0, 32, 32, 64, " in " ".join(l["kernel"] for l in m.launch_info())


# ------------------------------------------------------------------------------------------------ sub-batches on two streams
def test_batch32_distinct_images_two_streams(pkg, dev):
    """BASELINE configs[2] with 32 DISTINCT images: the two-stream forward (two staggered sub-batches of 16) is bit-identical
    to the one-stream forward, run-to-run deterministic, and ALL 32 images match the CPU port (tolerance 1e-4 on |y| <= ~32)."""
    res, seed = 512, 52
    m, sd = _model(pkg, res, seed, dev)
    x = pkg.synth.make_input(32, res, seed=seed, kind="demo")
    xt = torch.from_numpy(x).to(dev)
    with torch.no_grad():
        m.set_streams(2)
        y2 = m(xt)
        y2b = m(xt)
        m.set_streams(1)
        y1 = m(xt)
    torch.cuda.synchronize()
    assert torch.equal(y2, y2b) and torch.equal(y2, y1)
    got = y2.cpu()
    worst = 0.0
    for i0 in range(0, 32, 8):                                  # (the port runs ~1 image/s on the host: eight at a time)
        want = torc.generator(x[i0:i0 + 8], sd, res)
        worst = max(worst, float((got[i0:i0 + 8] - want).abs().max()))
    assert worst <= TOL, worst
    # the caller's stream is ordered after both sub-batches: work enqueued behind the forward sees the complete output
    with torch.no_grad():
        m.set_streams(2)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            ys = m(xt)
            total = ys.double().sum()
        s.synchronize()
    assert float(total) == float(y1.double().sum())


@pytest.mark.parametrize("res,storage", [(256, "bf16"), (512, "bf16"), (512, "f16"), (256, "f32")])
def test_two_stream_forward_is_deterministic_in_every_storage_mode(pkg, dev, res, storage):
    """32 distinct images, two staggered sub-batches on two streams, several runs: bit-identical run to run and identical to
    the one-stream result.  (This is the test that caught the packed-fp32 hazard in the ToRGB dot products of the 16-bit
    modes, profiles/r02_torgb_packed_f32_hazard.md: it only showed while kernels of the other stream shared the GPU.)"""
    seed = 54
    m, sd = _model(pkg, res, seed, dev, activation_dtype=storage)
    xt = torch.from_numpy(pkg.synth.make_input(32, res, seed=seed, kind="demo")).to(dev)
    with torch.no_grad():
        m.set_streams(1)
        y1 = m(xt).clone()
        m.set_streams(2)
        runs = [m(xt).clone() for _ in range(6)]
    torch.cuda.synchronize()
    for i, y in enumerate(runs):
        assert torch.equal(y, y1), (i, int((y != y1).sum()))


def test_ragged_sub_batches_256(pkg, dev):
    res, seed = 256, 53
    m, sd = _model(pkg, res, seed, dev)
    xt = torch.from_numpy(pkg.synth.make_input(21, res, seed=seed)).to(dev)
    with torch.no_grad():
        y2 = m(xt)
        m.set_streams(1)
        y1 = m(xt)
    assert torch.equal(y1, y2)


# ------------------------------------------------------------------------------------------------ static weights
def test_freeze_weights_contract(pkg, dev):
    res = 64
    m, sd = _model(pkg, res, 83, dev)
    x = torch.from_numpy(pkg.synth.make_input(2, res, seed=83)).to(dev)
    with torch.inference_mode():
        y0 = m(x).clone()
        m.synthesis.b64.conv2.conv2.weight.data.mul_(0.5)             # .data write: no version counter moves
        y1 = m(x).clone()                                              # default: always seen
        assert float((y1 - y0).abs().max()) > 1e-3
        m.freeze_weights()
        assert torch.equal(m(x), y1) and torch.equal(m(x), y1)
        m.synthesis.b64.conv2.conv2.weight.data.mul_(2.0)
        assert torch.equal(m(x), y1)                                   # frozen: not seen (documented)
        m.freeze_weights()                                             # documented way to invalidate
        y2 = m(x).clone()
        np.testing.assert_allclose(y2.cpu().numpy(), y0.cpu().numpy(), rtol=0, atol=3e-5 * float(y0.abs().max()))
        sd2 = pkg.synth.make_state_dict(res, seed=84)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd2.items()})      # re-binding is picked up while frozen
        y3 = m(x).cpu().numpy()
    np.testing.assert_allclose(y3, orc.generator(x.cpu().numpy(), sd2, res), rtol=0, atol=TOL)


# ------------------------------------------------------------------------------------------------ arbitrary-size forward (N4)
@pytest.mark.parametrize("res,hw,storage", [(64, (48, 80), "f32"), (64, (16, 16), "f32"), (256, (192, 320), "f32"), (256, (320, 128), "bf16"),
                                            (512, (384, 640), "f32")])
def test_forward_any_size_vs_oracle(pkg, dev, res, hw, storage):
    hh, ww = hw
    seed, batch = 91, 2
    m, sd = _model(pkg, res, seed, dev, activation_dtype=storage)
    x = (pkg.synth.normal((batch, 4, hh, ww), seed, "xhw") * 0.7).astype(np.float32)
    with torch.no_grad():
        y = m.forward_any_size(torch.from_numpy(x).to(dev)).cpu()
    want = torc.generator(x, sd, res, storage=storage)
    assert tuple(y.shape) == (batch, 3, hh, ww) and bool(torch.isfinite(y).all())
    if storage == "f32":
        assert float((y - want).abs().max()) <= TOL
    else:
        err_mode = float((want - torc.generator(x, sd, res)).abs().max())
        assert float((y - want).abs().max()) <= 2.0 * err_mode
    with pytest.raises(ValueError, match="multiples"):
        m.forward_any_size(torch.zeros(1, 4, hh + 1, ww, device=dev))


def test_forward_any_size_matches_reference_goldens(pkg, dev, golden_dir):
    """Outputs of the REFERENCE module run with its fixed-size buffers replaced by dynamic ones (tests/golden/make_golden_hw.py)."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(golden_dir, "generator_hw_*.npz")))
    assert len(files) >= 3
    for f in files:
        g = np.load(f)
        r, n, seed, hh, ww = int(g["resolution"]), int(g["batch"]), int(g["seed"]), int(g["height"]), int(g["width"])
        m, _ = _model(pkg, r, seed, dev)
        x = (pkg.synth.normal((n, 4, hh, ww), seed, "xhw") * 0.7).astype(np.float32)
        with torch.no_grad():
            y = m.forward_any_size(torch.from_numpy(x).to(dev)).cpu().numpy()
        np.testing.assert_allclose(y, g["y"], rtol=0, atol=3e-5 * max(1.0, float(g["y_absmax"])), err_msg=os.path.basename(f))


# ------------------------------------------------------------------------------------------------ uint8 in / uint8 out (N2)
@pytest.mark.parametrize("res,batch", [(256, 3), (512, 17), (64, 2)])
def test_forward_uint8_is_the_three_step_path(pkg, dev, res, batch):
    seed = 97
    m, sd = _model(pkg, res, seed, dev)
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(batch, res, res, 3), dtype=np.uint8)
    mask = np.where(rng.random((batch, res, res)) < 0.4, 0, 255).astype(np.uint8)
    mask[0, :3, :5] = 128
    it, mt = torch.from_numpy(img).to(dev), torch.from_numpy(mask).to(dev)
    with torch.no_grad():
        x = pkg.pipeline.preprocess(it, mt)
        y = m(x)
        want = pkg.pipeline.compose(y, it, mt)
        got = m.forward_uint8(it, mt)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    np.testing.assert_array_equal(x.cpu().numpy(), pp.preprocess(img, mask))
    np.testing.assert_array_equal(want.cpu().numpy(), pp.compose(y.cpu().numpy(), img, mask))
    assert bool((got[mt == 255] == it[mt == 255]).all())


# ------------------------------------------------------------------------------------------------ non-finite values
def test_finite_inputs_give_finite_outputs_and_nan_policy(pkg, dev):
    """Inputs far outside the training range saturate at the +-256 clamp of lrelu_agc (reference :21-23) and stay finite, like the
    reference's (Tensor.clamp).  A NaN input pixel does what it does in the reference module -- it propagates (tests/test_gpu_robust.py
    holds the mask to the oracle's); with the opt-in nan_policy="clamp" build the result is finite and deterministic."""
    res = 64
    m, sd = _model(pkg, res, 61, dev)
    x = pkg.synth.make_input(2, res, seed=61, kind="randn") * np.float32(1e6)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev))
        assert bool(torch.isfinite(y).all())
        xn = torch.from_numpy(pkg.synth.make_input(1, res, seed=61)).to(dev)
        xn[0, 1, 10, 10] = float("nan")
        assert bool(torch.isnan(m(xn)).all())
        mk, _ = _model(pkg, res, 61, dev, nan_policy="clamp")
        yn = mk(xn)
        assert bool(torch.isfinite(yn).all()) and torch.equal(yn, mk(xn))


# ------------------------------------------------------------------------------------------------ second oracle (N3)
@pytest.mark.parametrize("tag", ["r64", "r256"])
def test_training_snapshot_to_hip_forward(pkg, dev, golden_dir, tag):
    """SURVEY 8c second oracle + 8f N3 on the GPU: seeded weights of the reference's TRAINING generator (migan.py, depthwise,
    re-parameterised x9, noise_mode='const'; output recorded by tests/golden/make_golden_training.py) -> mi-gan_amd/convert.py
    -> load_state_dict -> HIP forward, within the north star's 1e-3 (the reference's own two models differ by 2.5e-5)."""
    import importlib
    from tests.test_convert import training_case
    conv = importlib.import_module("mi-gan_amd.convert")
    g, res, seed, train = training_case(pkg, golden_dir, tag)
    m = pkg.Generator(resolution=res)
    m.load_state_dict(conv.convert_training_state_dict(train, res), strict=True)
    m = m.to(dev).eval()
    x = torch.from_numpy(pkg.synth.make_input(int(g["batch"]), res, seed=seed)).to(dev)
    with torch.no_grad():
        y = m(x).cpu().numpy()
    err = float(np.abs(y - g["y_train"]).max())
    assert err <= 1e-4 * max(1.0, float(g["y_absmax"])), err
    assert err <= 1e-3


# ------------------------------------------------------------------------------------------------ FIR-up layers on the 256-column tile (round 4)
WIDE_UP = "migan::sepconv_wide_kernel<false, 0, false, true, true, true>"


@pytest.mark.parametrize("case", [
    dict(cin=64, cout=256, h=16, batch=2, noise=True, skip=True),
    dict(cin=96, cout=512, h=12, w=20, batch=3, noise=True),
    dict(cin=512, cout=256, h=64, batch=2, noise=True, skip=True),           # synthesis.b128.conv1 of migan-512
    dict(cin=512, cout=512, h=32, batch=4, noise=True, skip=True),           # synthesis.b64.conv1
    dict(cin=160, cout=256, h=6, w=14, batch=1, skip=True),
])
def test_fir_up_on_the_wide_tile(pkg, dev, tuned, case):
    """up=2 SeparableConv2d with Cout % 256 == 0: sepconv_wide_kernel<..., UP> against the oracle; with wide_up = 0 the 128-column kernel"""
    lib = pkg.load_library()
    run_sepconv_case(lib, pkg, CudaMem(dev), storage="f32", gemm=2, up=2, **case)
    assert lib.last_kernel() == WIDE_UP, lib.last_kernel()
    tuned("wide_up", 0)
    run_sepconv_case(lib, pkg, CudaMem(dev), storage="f32", gemm=2, up=2, **case)
    assert lib.last_kernel().startswith("migan::sepconv_kernel<2, "), lib.last_kernel()
