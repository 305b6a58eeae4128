"""keaki_amd -- MI355X (gfx950) backend for the BN254 hot path of brech1/keaki.

Only what the path needs lives here:
  csrc/      hand-written HIP kernels + the C ABI (include/keaki_hip.h) -> libkeaki_hip.so
  hip.py     ctypes binding of the C ABI (numpy arrays / raw device pointers in, no torch types)
  host/      C++ mirror of keaki's public API (kzg / kem / enc / vec) above the C ABI -> libkeaki_host.so
  keaki.py   ctypes binding of that mirror, used by the tests and the Laconic-OT harness

There is no CPU fallback: importing is cheap, but every operation needs libkeaki_hip.so and a
gfx950 device and raises KeakiHipError otherwise.
"""
from .hip import KeakiHip, KeakiHipError, lib_path, load_library  # noqa: F401

__all__ = ["KeakiHip", "KeakiHipError", "lib_path", "load_library"]
