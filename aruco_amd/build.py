"""Builds aruco_amd/libarucohip.so in-tree with hipcc (cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = os.path.join(_HERE, "libarucohip.so")


def library_path():
    return _LIB


def _stale():
    if not os.path.exists(_LIB):
        return True
    t = os.path.getmtime(_LIB)
    deps = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "arucohip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source into libarucohip.so. Raises if hipcc fails."""
    if not force and not _stale():
        return _LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found at %s — the HIP extension cannot be built" % hipcc)
    cmd = ["make", "-C", _CSRC, "-j6", "HIPCC=" + hipcc]
    if force:
        cmd.append("-B")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("building libarucohip.so failed")
    return _LIB
