"""aruco_amd — MI355X-native ArUco marker detection (the MarkerDetector::detect / BoardDetector hot path).

The product is the C-ABI library aruco_amd/libarucohip.so (HIP kernels for gfx950, include/arucohip.h) and the
header-only C++ shim include/aruco_hip_shim.hpp. This Python package is the host-side mirror used by the tests and
bench.py: a ctypes binding (capi), MarkerDetector / BoardDetector classes with the reference's method names
(detector), the synthetic stream generator (synth) and the frame-sharded multi-GPU driver (dist).
"""
from .build import build_library, library_path  # noqa: F401

__all__ = ["build_library", "library_path"]
