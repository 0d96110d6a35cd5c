"""Writes the ippm_config image of a test configuration (oracle/configs.py names: small, c2, ...) for tools/probe/abi_asan_smoke.cpp:
    python tools/write_config_image.py small cfg.bin"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
from configs import make_params  # noqa: E402
from ippmarl import _ffi  # noqa: E402
from ippmarl.derived import DerivedConstants  # noqa: E402

cfg = _ffi.make_config(DerivedConstants(make_params(sys.argv[1] if len(sys.argv) > 1 else "small")))
with open(sys.argv[2] if len(sys.argv) > 2 else "cfg.bin", "wb") as f:
    f.write(bytes(cfg))
print(C.sizeof(cfg), "bytes")
