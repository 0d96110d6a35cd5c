"""ippmarl -- MI355X-native implementation of the ipp-marl hot path (multi-UAV env step + COMA loop).

Python host code drives hand-written gfx950 HIP kernels (``lib/libippmarl.so``) through the C-ABI declared
in ``include/ippmarl.h``.  There is no CPU fallback: importing the engine without the built library raises.
"""
from .params import load_params, default_params  # noqa: F401
from .derived import DerivedConstants  # noqa: F401

__version__ = "0.1.0"
