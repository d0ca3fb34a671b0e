"""mi-gan_amd: MI355X-native (gfx950) MI-GAN generator forward.

Drop-in for ``lib.model_zoo.migan_inference`` of Picsart-AI-Research/MI-GAN:
``Generator(resolution)`` keeps the reference constructor, ``state_dict`` schema
and ``forward(x)`` contract, and runs the whole encoder/decoder as hand-written
HIP kernels behind a C ABI (include/migan_hip.h, libmigan_hip.so).  ``comodgan`` is the same for
``lib.model_zoo.comodgan`` (Co-Mod-GAN generator, include/comodgan_hip.h).

The directory name contains a dash, so import it with
``importlib.import_module("mi-gan_amd")`` or through the ``migan_amd`` alias
module at the repository root.
"""
from . import comodgan, comodgan_schema, convert, distributed, hipbind, pipeline, schema, synth  # noqa: F401
from .migan_inference import Generator  # noqa: F401
from .hipbind import MiganLib, MiganError, load_library, library_path  # noqa: F401



def install_into_reference(migan: bool = True, comodgan_too: bool = True) -> None:
    """Make the reference's own scripts pick up the MI355X modules without editing them: registers this package's modules as
    ``lib.model_zoo.migan_inference`` / ``lib.model_zoo.comodgan`` in ``sys.modules`` (call it before the script's imports, with
    the reference repository on ``sys.path``).  Covered (tests/test_demo_script.py, tests/test_convert.py): scripts/demo.py
    unmodified, end to end; the forward paths of scripts/evaluate_fid_lpips.py; ``copy_weights()`` of
    scripts/export_inference_model.py with this package's ``Generator`` as its destination.  Not covered: anything that traces
    or differentiates the module (ONNX / torch.jit export, training).  When the reference's model registry is importable the
    Co-Mod-GAN drop-in classes are also registered under the reference's names (``comodgan_mapping`` ... ``comodgan_generator``),
    which the replaced ``lib.model_zoo.comodgan`` would otherwise no longer provide to ``get_model()``.  See INTEGRATION.md 3."""
    import sys
    from . import comodgan as _cm, migan_inference as _mi
    if migan:
        sys.modules["lib.model_zoo.migan_inference"] = _mi
    if comodgan_too:
        sys.modules["lib.model_zoo.comodgan"] = _cm
        try:
            from lib.model_zoo.common.get_model import get_model     # the reference's registry (needs its lib/ on sys.path)
            reg = get_model()
            for name, cls in (("comodgan_mapping", _cm.Mapping), ("comodgan_encoder", _cm.Encoder),
                              ("comodgan_synthesis", _cm.Synthesis), ("comodgan_generator", _cm.Generator)):
                reg.register(cls, name, _cm.version)
        except Exception:
            pass


__all__ = ["install_into_reference", "Generator", "MiganLib", "MiganError", "load_library", "library_path", "schema", "synth", "pipeline", "convert", "comodgan"]
