"""Importable alias for the ``mi-gan_amd`` package (a dash is not a valid
identifier): ``import migan_amd`` == ``importlib.import_module("mi-gan_amd")``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("mi-gan_amd")
sys.modules[__name__] = _pkg
