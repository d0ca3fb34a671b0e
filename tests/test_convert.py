"""Checkpoint converter against the inference state_dict the reference's own copy_weights()
(scripts/export_inference_model.py:17-85) produced from the same training tensors (tests/golden/convert_r16.npz,
tests/golden/make_golden_convert.py).  The training tensors are regenerated from mi-gan_amd/synth.py by the
recipe stored in the fixture; the fixture holds expected outputs only."""
import importlib
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def case(pkg, golden_dir):
    g = np.load(os.path.join(golden_dir, "convert_r16.npz"))
    seed, res = int(g["seed"]), int(g["resolution"])
    # shapes of the training tensors follow from the inference schema
    shapes = {e.name: e.shape for e in pkg.schema.entries(res)}
    train = {}
    for key in [str(k) for k in g["recipe"]]:
        prefix, leaf = key.rsplit(".", 1)
        if leaf == "bias":
            shp, scale = (shapes[prefix + ".weight"][0],), 0.1
        elif leaf == "noise_const":
            shp, scale = shapes[prefix.rsplit(".", 1)[0] + ".noise_const"], 1.0
        elif leaf == "noise_strength":
            shp, scale = (1,), 0.3
        else:                                   # weight / w0..wn
            shp, scale = shapes[prefix + ".weight"], 1.0
        t = torch.from_numpy((pkg.synth.normal(tuple(shp), seed, key) * scale).astype(np.float32))
        train[key] = t.reshape(()) if leaf == "noise_strength" else t
    return g, res, train


def test_matches_the_reference_copy_weights(pkg, case):
    g, res, train = case
    conv = importlib.import_module("mi-gan_amd.convert")
    sd = conv.convert_training_state_dict(train, res)
    names = [e.name for e in pkg.schema.entries(res)]
    assert list(sd.keys()) == names
    checked = 0
    for k in names:
        got = sd[k].numpy()
        if "full/" + k in g:
            np.testing.assert_allclose(got, g["full/" + k], rtol=1e-6, atol=1e-7, err_msg=k)
        else:
            np.testing.assert_allclose(got.reshape(-1)[::97], g["samp/" + k], rtol=1e-6, atol=1e-7, err_msg=k)
            s1, s2 = g["stat/" + k]
            f = got.reshape(-1).astype(np.float64)
            assert abs(f.sum() - s1) <= 1e-6 * max(1.0, abs(s1)) + 1e-4 and abs((f ** 2).sum() - s2) <= 1e-5 * s2, k
        checked += 1
    assert checked == len(names)
    # every converted conv weight has unit L2 norm per output channel (reference :26)
    w = sd["encoder.b16.conv1.conv2.weight"]
    assert torch.allclose(w.square().sum(dim=[1, 2, 3]), torch.ones(w.shape[0]), atol=1e-5)


def test_loads_into_the_generator_and_rejects_incomplete_checkpoints(pkg, case):
    _, res, train = case
    conv = importlib.import_module("mi-gan_amd.convert")
    sd = conv.convert_training_state_dict(train, res)
    m = pkg.Generator(res)
    m.load_state_dict(sd)                                       # strict: keys and shapes line up
    broken = dict(train)
    del broken["synthesis.b8.conv1.conv2.noise_const"]
    with pytest.raises(KeyError):
        conv.convert_training_state_dict(broken, res)
    broken = {k: v for k, v in train.items() if not k.startswith("encoder.b16.conv1.conv1.")}
    with pytest.raises(KeyError):
        conv.convert_training_state_dict(broken, res)


# ------------------------------------------------------------------------------------------------------------------------
# Second, independent oracle: the reference's TRAINING generator (lib/model_zoo/migan.py, depthwise +
# re-parameterised, 9 tensors per conv, noise_mode='const') on seeded weights, and the reference's own copy_weights() of that
# very module tree (tests/golden/make_golden_training.py).  Here: regenerate the training tensors from the recipe, convert
# them with mi-gan_amd/convert.py, run OUR forward.
def training_case(pkg, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"training_{tag}.npz"))
    seed, res = int(g["seed"]), int(g["resolution"])
    train = {}
    for key, shp in zip([str(k) for k in g["keys"]], [str(s) for s in g["shapes"]]):
        shape = tuple(int(v) for v in shp.split(",")) if shp else ()
        leaf = key.rsplit(".", 1)[1]
        scale = 0.5 if leaf == "bias" else (0.3 if leaf == "noise_strength" else 1.0)
        t = torch.from_numpy((pkg.synth.normal(shape or (1,), seed, "train/" + key) * scale).astype(np.float32)).reshape(shape)
        train[key] = t
    return g, res, seed, train


@pytest.mark.parametrize("tag", ["r16", "r64", "r256"])
def test_training_snapshot_to_inference_forward(pkg, golden_dir, tag):
    from oracle import migan_torch_cpu as torc
    conv = importlib.import_module("mi-gan_amd.convert")
    g, res, seed, train = training_case(pkg, golden_dir, tag)
    sd = conv.convert_training_state_dict(train, res)
    # (a) the converted tensors are the ones the reference's copy_weights() wrote into the reference inference module
    for k, v in sd.items():
        s1, s2 = g["inf/" + k]
        f = v.reshape(-1).double()
        assert abs(float(f.sum()) - s1) <= 1e-5 * max(1.0, abs(s1), float(f.abs().sum())), k
        assert abs(float((f * f).sum()) - s2) <= 1e-5 * max(1.0, s2), k
    # (b) the inference forward on them reproduces the TRAINING generator's output (reference self-check: isclose(rtol=1e-3),
    # export_inference_model.py:149-151; measured 5e-6 .. 2.5e-5 between the two reference models)
    x = pkg.synth.make_input(int(g["batch"]), res, seed=seed)
    y = torc.generator(x, {k: v.numpy() for k, v in sd.items()}, res).numpy()
    assert float(np.abs(y - g["y_train"]).max()) <= 1e-4 * max(1.0, float(g["y_absmax"]))
    assert float(np.abs(y - g["y_inference"]).max()) <= 3e-5 * max(1.0, float(g["y_absmax"]))
    assert float(g["y_absmax"]) > 5.0


def test_training_snapshot_through_the_product_kernels(pkg, golden_dir):
    """the same chain with the forward executed by the product kernel source (CPU fiber emulator, C ABI)"""
    from tests.emu_util import aligned, emu_lib
    conv = importlib.import_module("mi-gan_amd.convert")
    g, res, seed, train = training_case(pkg, golden_dir, "r16")
    sd = conv.convert_training_state_dict(train, res)
    lib = emu_lib()
    h = pkg.hipbind.MiganHandle(lib, res)
    keep = {k: aligned(v.numpy().reshape(1) if v.ndim == 0 else v.numpy()) for k, v in sd.items()}
    for name, shape, _ in h.weights():
        h.set_weight(name, keep[name].ctypes.data, shape)
    h.commit()
    n = int(g["batch"])
    x = aligned(pkg.synth.make_input(n, res, seed=seed))
    y = aligned(np.full((n, 3, res, res), np.nan, np.float32))
    need = h.workspace_bytes(n)
    ws = np.zeros(need // 4 + 64, np.float32)
    h.forward(x.ctypes.data, y.ctypes.data, n, ws.ctypes.data, need)
    assert float(np.abs(y - g["y_train"]).max()) <= 1e-4 * max(1.0, float(g["y_absmax"]))


@pytest.mark.skipif(not os.path.isdir("/root/reference/scripts"), reason="the reference repository is only present in the build container")
def test_reference_copy_weights_accepts_our_generator_as_destination(pkg):
    """scripts/export_inference_model.py::copy_weights (:17-85), the reference's own function, run with the reference's TRAINING
    generator as source and THIS package's Generator as destination: every attribute it reads exists on our module tree
    (fromrgb None, conv2.bias None, use_noise), and the state_dict it leaves equals mi-gan_amd/convert.py's."""
    import sys
    import types
    import importlib.util
    conv = importlib.import_module("mi-gan_amd.convert")
    ref = "/root/reference"
    saved_path, saved = list(sys.path), {k: sys.modules.get(k) for k in ("cv2", "torchvision", "torchvision.transforms")}
    sys.path.insert(0, ref)
    try:
        for name in ("cv2", "torchvision", "torchvision.transforms"):
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
        if not hasattr(sys.modules["torchvision"], "transforms"):
            sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
        import lib.model_zoo.migan as train_mod
        spec = importlib.util.spec_from_file_location("ref_export_t", os.path.join(ref, "scripts", "export_inference_model.py"))
        ref_export = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref_export)
        res = 16
        kw = dict(resolution=res, ch_base=32768, ch_max=512, depthwise=True, reparametrize=True, num_reparam_tensors=9)
        G = train_mod.Generator(train_mod.Encoder(ic_n=4, **kw), train_mod.Synthesis(rgb_n=3, **kw)).eval()
        new = {}
        for k, v in G.state_dict().items():
            if "filter" in k.rsplit(".", 1)[1]:
                new[k] = v
            else:
                new[k] = torch.from_numpy((pkg.synth.normal(tuple(v.shape) or (1,), 5, "t/" + k) * 0.7).astype(np.float32)).reshape(v.shape)
        G.load_state_dict(new, strict=True)
        dest = pkg.Generator(resolution=res)
        with torch.no_grad():
            ref_export.copy_weights(G, dest, resolution=res)
        want = conv.convert_training_state_dict({k: v for k, v in new.items()}, res)
        got = dest.state_dict()
        assert list(got.keys()) == list(want.keys())
        for k in want:
            assert torch.allclose(got[k], want[k], rtol=1e-6, atol=1e-7), k
    finally:
        sys.path[:] = saved_path
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
        for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.") or k.startswith("torch_utils") or k.startswith("dnnlib")]:
            sys.modules.pop(k, None)
