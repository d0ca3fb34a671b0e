"""The reference's scripts/demo.py, UNMODIFIED, driving this package (build container only: needs /root/reference).

  * `install_into_reference()` registers our modules as lib.model_zoo.migan_inference / lib.model_zoo.comodgan, the real
    script file is executed (argparse, model selection :89-108, load_state_dict(torch.load(path)) :110, image / mask reading,
    resize, preprocess :122-130, the forward call :133-134, post-processing, composition, PNG output :135-142).
  * cv2 is not installed here: a stand-in provides the one function demo.py uses (cv2.resize, :138) on top of PIL.
  * There is no GPU here and the package has no CPU path: with --device cpu the script must reach the forward call and fail
    THERE with our "needs an MI355X" error (no silent fallback).
  * To also execute the forward, a second test points the module at the CPU fiber emulator of the product's kernel source
    (tests/emu; host memory plays the device) -- test plumbing only: three attributes are patched on the instance's class for
    the duration of the test.  The PNG demo.py writes is compared with the one the REFERENCE module produces through the same
    script.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch
from PIL import Image

REF = os.environ.get("MIGAN_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "scripts")), reason="the reference repository is only present in the build container")


def _cv2_stand_in():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_CUBIC = 2

    def resize(arr, dsize, interpolation=None):
        return np.array(Image.fromarray(arr).resize(tuple(dsize), Image.BICUBIC))

    cv2.resize = resize
    return cv2


def _load_demo(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "scripts", "demo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture()
def workdir(pkg, tmp_path):
    res = 256
    sd = pkg.synth.make_state_dict(res, seed=71)
    ckpt = tmp_path / "migan_256.pt"
    torch.save({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, ckpt)
    (tmp_path / "images").mkdir()
    (tmp_path / "masks").mkdir()
    ex = os.path.join(REF, "examples", "places2_512_object")
    for sub in ("images", "masks"):
        Image.open(os.path.join(ex, sub, "1.png")).save(tmp_path / sub / "1.png")           # 512 x 343 original
    return tmp_path, ckpt, sd


def _run_demo(demo, tmp, ckpt, out, device="cpu"):
    argv = sys.argv
    sys.argv = ["demo.py", "--model-name", "migan-256", "--model-path", str(ckpt), "--images-dir", str(tmp / "images"),
                "--masks-dir", str(tmp / "masks"), "--output-dir", str(out), "--device", device]
    try:
        demo.main()
    finally:
        sys.argv = argv


@pytest.fixture()
def patched_modules(pkg):
    saved = {k: sys.modules.get(k) for k in ("cv2", "lib.model_zoo.migan_inference", "lib.model_zoo.comodgan")}
    path = list(sys.path)
    sys.path.insert(0, REF)
    sys.modules["cv2"] = _cv2_stand_in()
    yield
    sys.path[:] = path
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_real_demo_script_reaches_our_forward_and_refuses_cpu(pkg, workdir, patched_modules):
    tmp, ckpt, _ = workdir
    pkg.install_into_reference()
    demo = _load_demo("ref_demo_ours")
    assert demo.MIGAN is pkg.Generator                                   # the script's own import line picked up the drop-in
    with pytest.raises(RuntimeError, match="MI355X"):
        _run_demo(demo, tmp, ckpt, tmp / "out_none")
    assert not list((tmp / "out_none").glob("*.png"))                    # nothing was written: no fallback produced an image


def test_real_demo_script_end_to_end_on_the_emulated_kernels(pkg, workdir, patched_modules, monkeypatch):
    from tests.emu_util import emu_lib
    tmp, ckpt, sd = workdir
    # expected picture: the same script with the REFERENCE module
    for k in ("lib.model_zoo.migan_inference", "lib.model_zoo.comodgan"):
        sys.modules.pop(k, None)
    demo_ref = _load_demo("ref_demo_reference")
    assert demo_ref.MIGAN.__module__ == "lib.model_zoo.migan_inference" and demo_ref.MIGAN is not pkg.Generator
    _run_demo(demo_ref, tmp, ckpt, tmp / "out_ref")
    # our module in its place, its C ABI calls served by the product kernel source on the fiber emulator
    pkg.install_into_reference()
    lib = emu_lib()
    G = pkg.Generator
    monkeypatch.setattr(G, "_require_device", lambda self, x: 0)
    monkeypatch.setattr(G, "_stream", lambda self, x: 0)
    monkeypatch.setattr(pkg.migan_inference, "load_library", lambda **kw: lib)
    demo = _load_demo("ref_demo_ours2")
    assert demo.MIGAN is G
    _run_demo(demo, tmp, ckpt, tmp / "out_ours")
    a = np.array(Image.open(tmp / "out_ref" / "1.png")).astype(np.int32)
    b = np.array(Image.open(tmp / "out_ours" / "1.png")).astype(np.int32)
    assert a.shape == b.shape == (171, 256, 3)                           # 512 x 343 original, resized by demo.resize to max 256
    d = np.abs(a - b)
    assert d.max() <= 2 and float((d > 0).mean()) < 2e-3                 # bicubic resize of uint8 images that differ by <= 1 count
