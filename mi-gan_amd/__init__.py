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

__all__ = ["Generator", "MiganLib", "MiganError", "load_library", "library_path", "schema", "synth", "pipeline", "convert", "comodgan"]
