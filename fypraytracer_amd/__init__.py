"""fypraytracer_amd — MI355X-native trace + shade path behind the FYPRayTracer Renderer API.

Only what the hot path needs: the ctypes face of the C ABI (`capi`), the host-side mirror of
the reference's Scene / Material / Camera producers (`scene`) and the synthetic scene
generators of the BASELINE configurations (`scenes`).  The HIP kernels and the C ABI live in
`csrc/` (libfyprt.so); the C++ facade with the reference's `Renderer` surface in `host/`.
"""
from . import capi, scene, scenes  # noqa: F401
from .capi import Context, FyprtError, Settings  # noqa: F401

__version__ = "0.1.0"
