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
    the reference repository on ``sys.path``; scripts/demo.py:15-21 then imports these classes).  See INTEGRATION.md section 3."""
    import sys
    from . import comodgan as _cm, migan_inference as _mi
    if migan:
        sys.modules["lib.model_zoo.migan_inference"] = _mi
    if comodgan_too:
        sys.modules["lib.model_zoo.comodgan"] = _cm


__all__ = ["install_into_reference", "Generator", "MiganLib", "MiganError", "load_library", "library_path", "schema", "synth", "pipeline", "convert", "comodgan"]
