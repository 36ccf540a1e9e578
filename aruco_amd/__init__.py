"""aruco_amd — MI355X-native ArUco marker detection (the MarkerDetector::detect / BoardDetector hot path).

The product is the C-ABI library aruco_amd/libarucohip.so (HIP kernels for gfx950, include/arucohip.h) and the
header-only C++ shim include/aruco_hip_shim.hpp (the reference's classes; the reference is C++, so is its host layer).
This Python package is what the tests and bench.py use: a ctypes binding of the C ABI (capi), the synthetic stream
generator (synth), the fixture readers (fixtures), the frame-sharded multi-GPU helpers for torch.distributed ranks (dist)
and the build script (build).
"""
from .build import build_library, library_path  # noqa: F401

__all__ = ["build_library", "library_path"]
