"""ctypes binding of include/arucohip.h (libarucohip.so). Fails loudly when the HIP library is missing —
there is no CPU fallback on the product path.

Import torch BEFORE this module when both are used in one process: torch bundles its own libamdhip64.so.7 and the
dynamic loader then resolves libarucohip's dependency to that already-loaded runtime (same SONAME), so device
pointers and streams can be shared between torch and the library.
"""
import ctypes as C
import os
import sys

import numpy as np

from .build import library_path

OK, E_INVALID, E_CAPACITY, E_UNSUPPORTED, E_HIP, E_OVERFLOW, E_BOARD_CONFIG = range(7)
THRES_FIXED, THRES_ADPT, THRES_CANNY = 0, 1, 2
CORNER_NONE, CORNER_HARRIS, CORNER_SUBPIX, CORNER_LINES = 0, 1, 2, 3
BOARD_NONE, BOARD_PIX, BOARD_METERS = -1, 0, 1

_ERR_NAMES = {1: "ARUCOHIP_E_INVALID", 2: "ARUCOHIP_E_CAPACITY", 3: "ARUCOHIP_E_UNSUPPORTED", 4: "ARUCOHIP_E_HIP",
              5: "ARUCOHIP_E_OVERFLOW", 6: "ARUCOHIP_E_BOARD_CONFIG"}


class ArucoHipError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s (%d): %s" % (_ERR_NAMES.get(code, "error"), code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("thres_method", C.c_int32), ("thres_param1_range", C.c_int32), ("thres_param1", C.c_double),
                ("thres_param2", C.c_double), ("corner_method", C.c_int32), ("warp_size", C.c_int32),
                ("min_size", C.c_float), ("max_size", C.c_float), ("border_dist", C.c_float),
                ("use_locked_corners", C.c_int32), ("decoder_kind", C.c_int32), ("erode", C.c_int32)]


class Marker(C.Structure):
    _fields_ = [("id", C.c_int32), ("corners", C.c_float * 8), ("ssize", C.c_float), ("has_pose", C.c_int32),
                ("pad_", C.c_int32), ("rvec", C.c_double * 3), ("tvec", C.c_double * 3)]


class BoardOut(C.Structure):
    _fields_ = [("n_markers", C.c_int32), ("has_pose", C.c_int32), ("rvec", C.c_double * 3), ("tvec", C.c_double * 3)]


class Limits(C.Structure):
    _fields_ = [("max_width", C.c_int32), ("max_height", C.c_int32), ("max_batch", C.c_int32),
                ("max_thres_planes", C.c_int32), ("triggers_per_frame", C.c_int32), ("contours_per_frame", C.c_int32),
                ("points_per_frame", C.c_int32), ("candidates_per_frame", C.c_int32), ("markers_per_frame", C.c_int32),
                ("long_walks_per_plane", C.c_int32)]


MARKER_DTYPE = np.dtype([("id", "<i4"), ("corners", "<f4", (8,)), ("ssize", "<f4"), ("has_pose", "<i4"),
                         ("pad_", "<i4"), ("rvec", "<f8", (3,)), ("tvec", "<f8", (3,))])
assert MARKER_DTYPE.itemsize == 96 and C.sizeof(Marker) == 96

# every symbol include/arucohip.h declares
SYMBOLS = [
    "arucohip_version", "arucohip_default_params", "arucohip_default_limits", "arucohip_create", "arucohip_create_ex",
    "arucohip_destroy", "arucohip_set_params", "arucohip_get_params", "arucohip_last_error_string", "arucohip_set_stream",
    "arucohip_get_stream", "arucohip_synchronize", "arucohip_detect", "arucohip_detect_batch", "arucohip_batch_status", "arucohip_batch_chunks",
    "arucohip_get_thresholded", "arucohip_get_candidates", "arucohip_threshold", "arucohip_detect_rectangles",
    "arucohip_warp", "arucohip_debug_num_contours", "arucohip_debug_contour", "arucohip_debug_candidates", "arucohip_debug_otsu",
    "arucohip_board_detect", "arucohip_calculate_extrinsics", "arucohip_stage_times", "arucohip_stage_name",
    "arucohip_enable_timing", "arucohip_kernel_times", "arucohip_threshold_exec_ms", "arucohip_kernel_name",
    "arucohip_debug_counters", "arucohip_board_detect_batch",
    "arucohip_gl_modelview", "arucohip_ogre_pose", "arucohip_gl_projection", "arucohip_ogre_projection",
    "arucohip_detect_bgr", "arucohip_detect_batch_bgr", "arucohip_bgr_to_gray", "arucohip_set_dictionary",
    "arucohip_set_decoder_callback",
    "arucohip_undistort", "arucohip_gl_modelview_n", "arucohip_gl_modelview_batch",
    "arucohip_set_pipeline_depth", "arucohip_detect_batch_submit", "arucohip_detect_batch_wait",
    "arucohip_mgpu_device_count", "arucohip_mgpu_create", "arucohip_mgpu_destroy", "arucohip_mgpu_size", "arucohip_mgpu_handle",
    "arucohip_mgpu_set_params", "arucohip_mgpu_last_error_string", "arucohip_mgpu_detect_batch", "arucohip_mgpu_detect_streams",
    "arucohip_mgpu_set_depth", "arucohip_mgpu_submit_batch", "arucohip_mgpu_submit_streams", "arucohip_mgpu_wait",
    "arucohip_compact_bytes", "arucohip_compact_markers", "arucohip_wait_event", "arucohip_detect_batch_retry_overflowed",
    "arucohip_refine_candidate_lines", "arucohip_mgpu_gather_mode", "arucohip_build_info",
]

_lib = None


def load():
    """dlopen libarucohip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # ARUCOHIP_LIB: an experiment's variant build (tools/ab_variants.sh, tools/stage_cost.sh) is loaded from its own path; the product
    # library in the tree is never overwritten
    path = os.environ.get("ARUCOHIP_LIB") or library_path()
    if "torch" not in sys.modules:
        # One HIP runtime per process: torch ships its own libamdhip64.so.7; when torch is importable load it first so
        # that libarucohip binds to the same runtime (two runtimes in one process cannot both own the GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise ArucoHipError(E_HIP, "libarucohip.so is not built (%s); run aruco_amd.build_library()" % path)
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    L.arucohip_last_error_string.restype = C.c_char_p
    L.arucohip_build_info.restype = C.c_char_p
    L.arucohip_stage_name.restype = C.c_char_p
    L.arucohip_kernel_name.restype = C.c_char_p
    L.arucohip_get_stream.restype = C.c_void_p
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.arucohip_create.argtypes = [vp, i, i, i, i, vp]
    L.arucohip_create_ex.argtypes = [vp, i, vp, vp]
    L.arucohip_destroy.argtypes = [vp]
    L.arucohip_set_params.argtypes = [vp, vp]
    L.arucohip_get_params.argtypes = [vp, vp]
    L.arucohip_last_error_string.argtypes = [vp]
    L.arucohip_set_stream.argtypes = [vp, vp]
    L.arucohip_get_stream.argtypes = [vp]
    L.arucohip_synchronize.argtypes = [vp]
    L.arucohip_detect.argtypes = [vp, vp, i, i, sz, vp, vp, i, f, i, vp, i, vp]
    L.arucohip_detect_batch.argtypes = [vp, vp, i, i, i, sz, sz, i, vp, vp, i, f, i, vp, i, vp, i]
    L.arucohip_batch_status.argtypes = [vp]
    L.arucohip_batch_chunks.argtypes = [vp, C.POINTER(C.c_int)]
    L.arucohip_get_thresholded.argtypes = [vp, i, vp]
    L.arucohip_get_candidates.argtypes = [vp, i, vp, i, vp]
    L.arucohip_threshold.argtypes = [vp, i, vp, i, i, sz, C.c_double, C.c_double, vp]
    L.arucohip_detect_rectangles.argtypes = [vp, vp, i, i, sz, vp, i, vp]
    L.arucohip_warp.argtypes = [vp, vp, i, i, sz, vp, i, vp]
    L.arucohip_debug_num_contours.argtypes = [vp, i, vp]
    L.arucohip_debug_contour.argtypes = [vp, i, i, vp, vp, vp, vp, i, vp]
    L.arucohip_debug_candidates.argtypes = [vp, i, vp, vp, vp, i, vp]
    L.arucohip_board_detect.argtypes = [vp, vp, i, vp, vp, i, i, vp, vp, i, f, f, i, vp, vp, vp]
    L.arucohip_calculate_extrinsics.argtypes = [vp, vp, i, vp, vp, i, f, i]
    L.arucohip_stage_times.argtypes = [vp, vp, i]
    L.arucohip_stage_name.argtypes = [i]
    L.arucohip_enable_timing.argtypes = [vp, i]
    L.arucohip_kernel_times.argtypes = [vp, vp, i]
    L.arucohip_threshold_exec_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.arucohip_kernel_name.argtypes = [i]
    L.arucohip_debug_counters.argtypes = [vp, vp]
    L.arucohip_board_detect_batch.argtypes = [vp, i, vp, vp, i, i, vp, vp, i, f, f, i, vp, vp]
    L.arucohip_detect_bgr.argtypes = [vp, vp, i, i, sz, vp, vp, i, f, i, vp, i, vp]
    L.arucohip_detect_batch_bgr.argtypes = [vp, vp, i, i, i, sz, sz, i, vp, vp, i, f, i, vp, i, vp, i]
    L.arucohip_bgr_to_gray.argtypes = [vp, vp, i, i, sz, vp]
    L.arucohip_set_dictionary.argtypes = [vp, i, i, vp, i, f]
    L.arucohip_undistort.argtypes = [vp, vp, i, i, i, sz, sz, i, i, vp, vp, i, vp, i]
    L.arucohip_gl_modelview.argtypes = [vp, vp, vp]
    L.arucohip_gl_modelview_n.argtypes = [vp, i, vp]
    L.arucohip_gl_modelview_batch.argtypes = [vp, i, i, vp, vp]
    L.arucohip_ogre_pose.argtypes = [vp, vp, vp, vp]
    L.arucohip_gl_projection.argtypes = [vp, i, i, i, i, C.c_double, C.c_double, i, vp]
    L.arucohip_ogre_projection.argtypes = [vp, i, i, i, i, C.c_double, C.c_double, i, vp]
    L.arucohip_set_decoder_callback.argtypes = [vp, vp, vp]
    L.arucohip_set_pipeline_depth.argtypes = [vp, i]
    L.arucohip_detect_batch_submit.argtypes = [vp, vp, i, i, i, sz, sz, i, vp, vp, i, f, i, vp, i, vp, i, vp]
    L.arucohip_detect_batch_wait.argtypes = [vp, i]
    L.arucohip_mgpu_create.argtypes = [vp, vp, i, i, i, i, i, i, vp]
    L.arucohip_mgpu_destroy.argtypes = [vp]
    L.arucohip_mgpu_destroy.restype = None
    L.arucohip_mgpu_size.argtypes = [vp]
    L.arucohip_mgpu_handle.argtypes = [vp, i]
    L.arucohip_mgpu_handle.restype = vp
    L.arucohip_mgpu_set_params.argtypes = [vp, vp]
    L.arucohip_mgpu_last_error_string.argtypes = [vp]
    L.arucohip_mgpu_last_error_string.restype = C.c_char_p
    L.arucohip_mgpu_detect_batch.argtypes = [vp, vp, i, i, i, sz, sz, vp, vp, i, f, i, vp, i, vp]
    L.arucohip_mgpu_detect_streams.argtypes = [vp, vp, vp, i, i, sz, sz, vp, vp, i, f, i, vp, i, vp]
    L.arucohip_mgpu_set_depth.argtypes = [vp, i]
    L.arucohip_mgpu_submit_batch.argtypes = [vp, vp, i, i, i, sz, sz, vp, vp, i, f, i, vp, i, vp, vp]
    L.arucohip_mgpu_submit_streams.argtypes = [vp, vp, vp, i, i, sz, sz, vp, vp, i, f, i, vp, i, vp, vp]
    L.arucohip_mgpu_wait.argtypes = [vp, i]
    L.arucohip_mgpu_gather_mode.argtypes = [vp]
    L.arucohip_compact_bytes.argtypes = [i, i]
    L.arucohip_compact_bytes.restype = sz
    L.arucohip_compact_markers.argtypes = [vp, vp, i, i, vp, i, vp]
    L.arucohip_wait_event.argtypes = [vp, vp]
    L.arucohip_detect_batch_retry_overflowed.argtypes = [vp, vp, i, i, i, sz, sz, i, vp, vp, i, f, i, vp, i, vp, i, vp]
    L.arucohip_refine_candidate_lines.argtypes = [vp, vp, i, vp, vp, vp, i]
    L.arucohip_default_params.argtypes = [vp]
    L.arucohip_default_limits.argtypes = [vp, i, i, i]
    _lib = L
    return L


def build_info():
    """arucohip_build_info(): 'src=<digest> flags=[...]' of the loaded library."""
    return (load().arucohip_build_info() or b"").decode()


def compact_bytes(nframes, cap_total):
    """Size of the packed gather block of arucohip_compact_markers."""
    return int(load().arucohip_compact_bytes(int(nframes), int(cap_total)))


def compact_markers(blocks_ptr, counts_ptr, nframes, cap, dst_ptr, cap_total, stream_ptr=0):
    """arucohip_compact_markers on device pointers (plain integers) and a HIP stream pointer."""
    rc = load().arucohip_compact_markers(C.c_void_p(blocks_ptr), C.c_void_p(counts_ptr), int(nframes), int(cap), C.c_void_p(dst_ptr), int(cap_total),
                                          C.c_void_p(stream_ptr))
    if rc != OK:
        raise ArucoHipError(rc, "arucohip_compact_markers")


def default_params():
    p = Params()
    load().arucohip_default_params(C.byref(p))
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class Handle:
    """Owns one arucohip_handle (one HIP stream + device buffers)."""

    def __init__(self, max_width, max_height, max_batch=1, device=0, params=None, limits=None):
        self.L = load()
        self.h = C.c_void_p()
        p = params if params is not None else default_params()
        if limits is None:
            rc = self.L.arucohip_create(C.byref(p), device, max_width, max_height, max_batch, C.byref(self.h))
        else:
            rc = self.L.arucohip_create_ex(C.byref(p), device, C.byref(limits), C.byref(self.h))
        if rc != OK:
            raise ArucoHipError(rc, "arucohip_create")
        self.max_batch = max_batch
        self.device = device

    def close(self):
        if self.h:
            self.L.arucohip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != OK and rc not in allow:
            raise ArucoHipError(rc, (self.L.arucohip_last_error_string(self.h) or b"").decode())
        return rc

    def get_params(self):
        p = Params()
        self._chk(self.L.arucohip_get_params(self.h, C.byref(p)))
        return p

    def set_params(self, p):
        self._chk(self.L.arucohip_set_params(self.h, C.byref(p)))

    def set_stream(self, stream_ptr):
        self._chk(self.L.arucohip_set_stream(self.h, C.c_void_p(stream_ptr)))

    def get_stream(self):
        return self.L.arucohip_get_stream(self.h)

    def synchronize(self):
        self._chk(self.L.arucohip_synchronize(self.h))

    def wait_event(self, event_ptr):
        """The handle's next work waits for a hipEvent_t (plain integer) recorded behind the producer of the frames."""
        self._chk(self.L.arucohip_wait_event(self.h, C.c_void_p(event_ptr)))

    # ---- host-buffer API
    def detect(self, gray, K=None, dist=None, marker_size=-1.0, y_perp=False, cap=128):
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = g.shape
        Ka, da = _f32(K), _f32(dist)
        out = np.zeros(cap, MARKER_DTYPE)
        n = C.c_int(0)
        self._chk(self.L.arucohip_detect(self.h, _ptr(g), w, h, w, _ptr(Ka), _ptr(da), 0 if da is None else da.size,
                                         float(marker_size), int(bool(y_perp)), _ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def set_dictionary(self, markers, tau0, rate=1.0):
        """HighlyReliableMarkers::loadDictionary: markers = bit strings of n*n characters; selects the HRM decoder
        (None / empty list: back to the 5x5 fiducial decoder)."""
        p = self.get_params()
        if not markers:
            p.decoder_kind = 0
            self.set_params(p)
            self._chk(self.L.arucohip_set_dictionary(self.h, 0, 0, None, 0, 1.0))
            return
        n = int(round(len(markers[0]) ** 0.5))
        codes = np.array([sum(1 << i for i, ch in enumerate(m) if ch == "1") for m in markers], np.uint64)
        self._chk(self.L.arucohip_set_dictionary(self.h, n, len(codes), _ptr(codes), int(tau0), float(rate)))
        p.decoder_kind = 1
        self.set_params(p)

    def detect_bgr(self, bgr, K=None, dist=None, marker_size=-1.0, y_perp=False, cap=128):
        """One host frame [H][W][3] in B,G,R order: converted to gray on the device, then detect()."""
        b = np.ascontiguousarray(bgr, dtype=np.uint8)
        h, w, c = b.shape
        assert c == 3
        Ka, da = _f32(K), _f32(dist)
        out = np.zeros(cap, MARKER_DTYPE)
        n = C.c_int(0)
        self._chk(self.L.arucohip_detect_bgr(self.h, _ptr(b), w, h, 3 * w, _ptr(Ka), _ptr(da), 0 if da is None else da.size,
                                             float(marker_size), int(bool(y_perp)), _ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def detect_batch_bgr_host(self, frames, K=None, dist=None, marker_size=-1.0, y_perp=False, cap=128):
        fr = np.ascontiguousarray(frames, dtype=np.uint8)
        nf, h, w, c = fr.shape
        assert c == 3
        Ka, da = _f32(K), _f32(dist)
        out = np.zeros((nf, cap), MARKER_DTYPE)
        n = np.zeros(nf, np.int32)
        self._chk(self.L.arucohip_detect_batch_bgr(self.h, _ptr(fr), nf, w, h, 3 * w, 3 * w * h, 0, _ptr(Ka), _ptr(da),
                                                   0 if da is None else da.size, float(marker_size), int(bool(y_perp)), _ptr(out), cap, _ptr(n), 0))
        return [out[f, :n[f]].copy() for f in range(nf)]

    def undistort(self, img, K, dist):
        """cv::undistort of host frames: [H][W], [H][W][3], [N][H][W] (gray batch) -> same shape."""
        a = np.ascontiguousarray(img, dtype=np.uint8)
        if a.ndim == 2:
            n, (hgt, wid), cn = 1, a.shape, 1
        elif a.ndim == 3 and a.shape[2] == 3:
            n, (hgt, wid), cn = 1, a.shape[:2], 3
        else:
            n, hgt, wid = a.shape
            cn = 1
        Ka, da = _f32(K), _f32(dist)
        out = np.empty_like(a)
        self._chk(self.L.arucohip_undistort(self.h, _ptr(a), n, wid, hgt, wid * cn, wid * hgt * cn, cn, 0, _ptr(Ka), _ptr(da),
                                            0 if da is None else da.size, _ptr(out), 0))
        return out

    def refine_candidate_lines(self, contour, corners, K=None, dist=None):
        """MarkerDetector::refineCandidateLines: contour = n x 2 integer points, corners = 4 x 2; returns the refined corners (4 x 2)."""
        xy = np.ascontiguousarray(contour, dtype=np.int32).reshape(-1, 2)
        c = np.ascontiguousarray(corners, dtype=np.float32).reshape(8).copy()
        Ka, da = _f32(K), _f32(dist)
        self._chk(self.L.arucohip_refine_candidate_lines(self.h, _ptr(xy), len(xy), _ptr(c), _ptr(Ka), _ptr(da), 0 if da is None else da.size))
        return c.reshape(4, 2)

    def bgr_to_gray(self, bgr):
        b = np.ascontiguousarray(bgr, dtype=np.uint8)
        h, w, c = b.shape
        assert c == 3
        g = np.empty((h, w), np.uint8)
        self._chk(self.L.arucohip_bgr_to_gray(self.h, _ptr(b), w, h, 3 * w, _ptr(g)))
        return g

    def detect_batch_host(self, frames, K=None, dist=None, marker_size=-1.0, y_perp=False, cap=128):
        fr = np.ascontiguousarray(frames, dtype=np.uint8)
        nf, h, w = fr.shape
        Ka, da = _f32(K), _f32(dist)
        out = np.zeros((nf, cap), MARKER_DTYPE)
        n = np.zeros(nf, np.int32)
        self._chk(self.L.arucohip_detect_batch(self.h, _ptr(fr), nf, w, h, w, w * h, 0, _ptr(Ka), _ptr(da),
                                               0 if da is None else da.size, float(marker_size), int(bool(y_perp)), _ptr(out),
                                               cap, _ptr(n), 0))
        return [out[f, :n[f]].copy() for f in range(nf)]

    # ---- device-pointer API (frames resident in HBM, results left in HBM): pointers are plain integers
    def detect_batch_device(self, frames_ptr, nframes, width, height, out_ptr, cap, n_out_ptr, K=None, dist=None,
                            marker_size=-1.0, y_perp=False, row_stride=None, frame_stride=None):
        Ka, da = _f32(K), _f32(dist)
        rs = width if row_stride is None else row_stride
        fs = rs * height if frame_stride is None else frame_stride
        self._chk(self.L.arucohip_detect_batch(self.h, C.c_void_p(frames_ptr), nframes, width, height, rs, fs, 1, _ptr(Ka),
                                               _ptr(da), 0 if da is None else da.size, float(marker_size), int(bool(y_perp)),
                                               C.c_void_p(out_ptr), cap, C.c_void_p(n_out_ptr), 1))

    def retry_overflowed_device(self, frames_ptr, nframes, width, height, out_ptr, cap, n_out_ptr, K=None, dist=None, marker_size=-1.0, y_perp=False):
        """arucohip_detect_batch_retry_overflowed on device frames / device results (after batch_status or wait returned E_OVERFLOW):
        returns the number of frames that were run again."""
        Ka, da = _f32(K), _f32(dist)
        k = C.c_int(0)
        self._chk(self.L.arucohip_detect_batch_retry_overflowed(self.h, C.c_void_p(frames_ptr), nframes, width, height, width, width * height, 1, _ptr(Ka),
                                                                _ptr(da), 0 if da is None else da.size, float(marker_size), int(bool(y_perp)),
                                                                C.c_void_p(out_ptr), cap, C.c_void_p(n_out_ptr), 1, C.byref(k)))
        return k.value

    def detect_batch_host_tolerant(self, frames, K=None, dist=None, marker_size=-1.0, y_perp=False, cap=128, retry=True):
        """detect_batch_host that survives list overflows: returns (per-frame arrays or None for a frame still given up, frames retried)."""
        fr = np.ascontiguousarray(frames, dtype=np.uint8)
        nf, h, w = fr.shape
        Ka, da = _f32(K), _f32(dist)
        out = np.zeros((nf, cap), MARKER_DTYPE)
        n = np.zeros(nf, np.int32)
        nd = 0 if da is None else da.size
        rc = self._chk(self.L.arucohip_detect_batch(self.h, _ptr(fr), nf, w, h, w, w * h, 0, _ptr(Ka), _ptr(da), nd, float(marker_size), int(bool(y_perp)),
                                                    _ptr(out), cap, _ptr(n), 0), allow=(E_OVERFLOW,))
        k = C.c_int(0)
        first = n.copy()
        if rc == E_OVERFLOW and retry:
            self._chk(self.L.arucohip_detect_batch_retry_overflowed(self.h, _ptr(fr), nf, w, h, w, w * h, 0, _ptr(Ka), _ptr(da), nd, float(marker_size),
                                                                    int(bool(y_perp)), _ptr(out), cap, _ptr(n), 0, C.byref(k)))
        return [out[f, :n[f]].copy() if n[f] >= 0 else None for f in range(nf)], k.value, first

    def detect_batch_mixed(self, frames_host_ptr, nframes, width, height, out_ptr, cap, n_out_ptr, K=None, dist=None,
                           marker_size=-1.0, y_perp=False):
        """Frames in (pinned) host memory, results left on the device: the PCIe-inclusive path."""
        Ka, da = _f32(K), _f32(dist)
        self._chk(self.L.arucohip_detect_batch(self.h, C.c_void_p(frames_host_ptr), nframes, width, height, width, width * height, 0,
                                               _ptr(Ka), _ptr(da), 0 if da is None else da.size, float(marker_size), int(bool(y_perp)),
                                               C.c_void_p(out_ptr), cap, C.c_void_p(n_out_ptr), 1))

    # ---- batches in flight
    def set_pipeline_depth(self, depth):
        self._chk(self.L.arucohip_set_pipeline_depth(self.h, int(depth)))

    def submit_device(self, frames_ptr, nframes, width, height, out_ptr, cap, n_out_ptr, K=None, dist=None, marker_size=-1.0, y_perp=False,
                      frames_on_device=True):
        """arucohip_detect_batch_submit with results left on the device (frames device-resident, or in pinned host memory with
        frames_on_device=False); returns the ticket."""
        Ka, da = _f32(K), _f32(dist)
        t = C.c_int(-1)
        self._chk(self.L.arucohip_detect_batch_submit(self.h, C.c_void_p(frames_ptr), nframes, width, height, width, width * height,
                                                      int(bool(frames_on_device)), _ptr(Ka),
                                                      _ptr(da), 0 if da is None else da.size, float(marker_size), int(bool(y_perp)),
                                                      C.c_void_p(out_ptr), cap, C.c_void_p(n_out_ptr), 1, C.byref(t)))
        return t.value

    def submit_host(self, frames, out, n, K=None, dist=None, marker_size=-1.0, y_perp=False):
        """Host frames [n][H][W] and host result arrays (out: [n][cap] MARKER_DTYPE, n: int32[n]) that must stay alive
        until wait(ticket)."""
        nf, h, w = frames.shape
        Ka, da = _f32(K), _f32(dist)
        t = C.c_int(-1)
        self._chk(self.L.arucohip_detect_batch_submit(self.h, _ptr(frames), nf, w, h, w, w * h, 0, _ptr(Ka), _ptr(da), 0 if da is None else da.size,
                                                      float(marker_size), int(bool(y_perp)), _ptr(out), out.shape[1], _ptr(n), 0, C.byref(t)))
        return t.value

    def wait(self, ticket, allow=()):
        return self._chk(self.L.arucohip_detect_batch_wait(self.h, int(ticket)), allow)

    def batch_status(self):
        return self._chk(self.L.arucohip_batch_status(self.h))

    def batch_chunks(self):
        """(chunks, frames per chunk) of the last batch: every kernel launch covers one chunk."""
        per = C.c_int(0)
        n = self.L.arucohip_batch_chunks(self.h, C.byref(per))
        return n, per.value

    def thresholded(self, frame=0, shape=None):
        out = np.empty(shape, np.uint8)
        self._chk(self.L.arucohip_get_thresholded(self.h, frame, _ptr(out)))
        return out

    def candidates(self, frame=0, cap=512):
        q = np.zeros((cap, 4, 2), np.float32)
        n = C.c_int(0)
        self._chk(self.L.arucohip_get_candidates(self.h, frame, _ptr(q), cap, C.byref(n)))
        return q[:n.value].copy()

    def debug_candidates(self, frame=0, cap=512):
        q = np.zeros((cap, 4, 2), np.float32)
        ids = np.zeros(cap, np.int32)
        nrot = np.zeros(cap, np.int32)
        n = C.c_int(0)
        self._chk(self.L.arucohip_debug_candidates(self.h, frame, _ptr(q), _ptr(ids), _ptr(nrot), cap, C.byref(n)))
        k = n.value
        return q[:k].copy(), ids[:k].copy(), nrot[:k].copy()

    def debug_otsu(self, frame=0, cap=512):
        """Otsu threshold of every candidate's patch (candidate order of debug_candidates)."""
        t = np.zeros(cap, np.int32)
        n = C.c_int(0)
        self._chk(self.L.arucohip_debug_otsu(self.h, frame, _ptr(t), cap, C.byref(n)))
        return t[:n.value].copy()

    def debug_contours(self, frame=0):
        n = C.c_int(0)
        self._chk(self.L.arucohip_debug_num_contours(self.h, frame, C.byref(n)))
        res = []
        for i in range(n.value):
            hole, sx, sy, npts = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            self._chk(self.L.arucohip_debug_contour(self.h, frame, i, C.byref(hole), C.byref(sx), C.byref(sy), None, 0, C.byref(npts)))
            pts = np.zeros((npts.value, 2), np.int16)
            self._chk(self.L.arucohip_debug_contour(self.h, frame, i, C.byref(hole), C.byref(sx), C.byref(sy), _ptr(pts),
                                                    npts.value, C.byref(npts)))
            res.append({"hole": hole.value, "start": (sx.value, sy.value), "pts": pts.astype(np.int32)})
        return res

    def threshold(self, gray, method=THRES_ADPT, param1=-1.0, param2=-1.0):
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = g.shape
        out = np.empty((h, w), np.uint8)
        self._chk(self.L.arucohip_threshold(self.h, method, _ptr(g), w, h, w, float(param1), float(param2), _ptr(out)))
        return out

    def detect_rectangles(self, thres, cap=512):
        t = np.ascontiguousarray(thres, dtype=np.uint8)
        h, w = t.shape
        q = np.zeros((cap, 4, 2), np.float32)
        n = C.c_int(0)
        self._chk(self.L.arucohip_detect_rectangles(self.h, _ptr(t), w, h, w, _ptr(q), cap, C.byref(n)))
        return q[:n.value].copy()

    def warp(self, gray, quad, size=56):
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = g.shape
        q = _f32(quad).reshape(8)
        out = np.empty((size, size), np.uint8)
        self._chk(self.L.arucohip_warp(self.h, _ptr(g), w, h, w, _ptr(q), size, _ptr(out)))
        return out

    def calculate_extrinsics(self, markers, K, dist, marker_size, y_perp=False):
        m = np.ascontiguousarray(markers, dtype=MARKER_DTYPE).copy()
        Ka, da = _f32(K), _f32(dist)
        self._chk(self.L.arucohip_calculate_extrinsics(self.h, _ptr(m), len(m), _ptr(Ka), _ptr(da), 0 if da is None else da.size,
                                                       float(marker_size), int(bool(y_perp))))
        return m

    def board_detect(self, markers, ids, obj, info_type, K=None, dist=None, marker_size=-1.0, repj_err_thres=-1.0, y_perp=False):
        m = np.ascontiguousarray(markers, dtype=MARKER_DTYPE)
        ida = np.ascontiguousarray(ids, dtype=np.int32)
        oa = _f32(obj)
        Ka, da = _f32(K), _f32(dist)
        outm = np.zeros(max(len(m), 1), MARKER_DTYPE)
        bo = BoardOut()
        prob = C.c_float(0)
        self._chk(self.L.arucohip_board_detect(self.h, _ptr(m) if len(m) else None, len(m), _ptr(ida), _ptr(oa), len(ida), info_type,
                                               _ptr(Ka), _ptr(da), 0 if da is None else da.size, float(marker_size),
                                               float(repj_err_thres), int(bool(y_perp)), _ptr(outm), C.byref(bo), C.byref(prob)))
        return {"prob": prob.value, "markers": outm[:bo.n_markers].copy(), "has_pose": bo.has_pose,
                "rvec": np.array(bo.rvec), "tvec": np.array(bo.tvec)}

    def enable_timing(self, on=True):
        self.L.arucohip_enable_timing(self.h, int(on))

    def stage_times(self):
        ms = (C.c_float * 8)()
        n = self.L.arucohip_stage_times(self.h, ms, 8)
        return {self.L.arucohip_stage_name(i).decode(): ms[i] for i in range(n)}

    def kernel_times(self):
        ms = (C.c_float * 16)()
        n = self.L.arucohip_kernel_times(self.h, ms, 16)
        return {self.L.arucohip_kernel_name(i).decode(): ms[i] for i in range(n)}

    def threshold_exec_ms(self):
        """(total ms, launches) of the wide threshold kernel since enable_timing(True), from the device clock stamps of its waves."""
        ms, n = C.c_double(0), C.c_int(0)
        self._chk(self.L.arucohip_threshold_exec_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def debug_counters(self):
        c = np.zeros(8, np.uint32)
        self._chk(self.L.arucohip_debug_counters(self.h, _ptr(c)))
        return {"raw": int(c[4]), "triggers": int(c[0]), "contours": int(c[1]), "points": int(c[2]), "status": int(c[3]),
                "long_walks": int(c[4])}   # walker mode: [4] = long walks (checkpoint rings handed out)

    def gl_modelview_batch(self, nframes, cap=64):
        """Marker::glGetModelViewMatrix for every marker of the last batch (device kernel): list per frame of [n][16]."""
        mv = np.zeros((nframes, cap, 16), np.float64)
        n = np.zeros(nframes, np.int32)
        self._chk(self.L.arucohip_gl_modelview_batch(self.h, nframes, cap, _ptr(mv), _ptr(n)))
        return [mv[f, :n[f]].copy() for f in range(nframes)]

    def board_detect_batch(self, nframes, ids, obj, info_type, K=None, dist=None, marker_size=-1.0, repj_err_thres=-1.0, y_perp=False):
        """BoardDetector::detect on the device-resident markers of the last detect_batch call, all frames at once."""
        ida = np.ascontiguousarray(ids, dtype=np.int32)
        oa = _f32(obj)
        Ka, da = _f32(K), _f32(dist)
        out = (BoardOut * nframes)()
        prob = np.zeros(nframes, np.float32)
        self._chk(self.L.arucohip_board_detect_batch(self.h, nframes, _ptr(ida), _ptr(oa), len(ida), info_type, _ptr(Ka), _ptr(da),
                                                     0 if da is None else da.size, float(marker_size), float(repj_err_thres),
                                                     int(bool(y_perp)), out, _ptr(prob)))
        return [{"n_markers": out[f].n_markers, "has_pose": out[f].has_pose, "rvec": np.array(out[f].rvec), "tvec": np.array(out[f].tvec),
                 "prob": float(prob[f])} for f in range(nframes)]


class MultiGpu:
    """arucohip_mgpu_*: frames sharded round-robin over device slots, marker blocks gathered (host or peer/xGMI)."""

    GATHER_HOST, GATHER_PEER = 0, 1

    def __init__(self, devices, max_width, max_height, frames_per_device, cap=64, flags=0, params=None):
        self.L = load()
        self.m = C.c_void_p()
        p = params if params is not None else default_params()
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        rc = self.L.arucohip_mgpu_create(C.byref(p), _ptr(dv), len(dv), max_width, max_height, frames_per_device, cap, flags, C.byref(self.m))
        if rc != OK:
            raise ArucoHipError(rc, "arucohip_mgpu_create")
        self.cap, self.per, self.G = cap, frames_per_device, len(dv)

    def close(self):
        if self.m:
            self.L.arucohip_mgpu_destroy(self.m)
            self.m = C.c_void_p()

    def _chk(self, rc):
        if rc != OK:
            raise ArucoHipError(rc, (self.L.arucohip_mgpu_last_error_string(self.m) or b"").decode())

    def detect_batch_host(self, frames, K=None, dist=None, marker_size=-1.0, y_perp=False):
        fr = np.ascontiguousarray(frames, dtype=np.uint8)
        nf, h, w = fr.shape
        Ka, da = _f32(K), _f32(dist)
        out = np.zeros((nf, self.cap), MARKER_DTYPE)
        n = np.zeros(nf, np.int32)
        self._chk(self.L.arucohip_mgpu_detect_batch(self.m, _ptr(fr), nf, w, h, w, w * h, _ptr(Ka), _ptr(da), 0 if da is None else da.size,
                                                    float(marker_size), int(bool(y_perp)), _ptr(out), self.cap, _ptr(n)))
        return [out[f, :n[f]].copy() for f in range(nf)]

    def set_depth(self, depth):
        self._chk(self.L.arucohip_mgpu_set_depth(self.m, int(depth)))

    def gather_mode(self):
        """GATHER_PEER only if it was asked for and every device reaches the first one; else GATHER_HOST."""
        return int(self.L.arucohip_mgpu_gather_mode(self.m))

    def submit_batch_host(self, frames, K=None, dist=None, marker_size=-1.0, y_perp=False):
        """arucohip_mgpu_submit_batch; returns a job whose arrays stay alive until wait(job)."""
        fr = np.ascontiguousarray(frames, dtype=np.uint8)
        nf, h, w = fr.shape
        Ka, da = _f32(K), _f32(dist)
        out = np.zeros((nf, self.cap), MARKER_DTYPE)
        n = np.zeros(nf, np.int32)
        t = C.c_int(-1)
        self._chk(self.L.arucohip_mgpu_submit_batch(self.m, _ptr(fr), nf, w, h, w, w * h, _ptr(Ka), _ptr(da), 0 if da is None else da.size,
                                                    float(marker_size), int(bool(y_perp)), _ptr(out), self.cap, _ptr(n), C.byref(t)))
        return {"ticket": t.value, "frames": fr, "out": out, "n": n, "K": Ka, "dist": da, "kind": "batch"}

    def submit_streams(self, ptrs, counts, width, height, K=None, dist=None, marker_size=-1.0, y_perp=False):
        pa = (C.c_void_p * self.G)(*[C.c_void_p(int(x)) for x in ptrs])
        ca = np.ascontiguousarray(counts, dtype=np.int32)
        Ka, da = _f32(K), _f32(dist)
        out = np.zeros((self.G * self.per, self.cap), MARKER_DTYPE)
        n = np.zeros(self.G * self.per, np.int32)
        t = C.c_int(-1)
        self._chk(self.L.arucohip_mgpu_submit_streams(self.m, pa, _ptr(ca), width, height, width, width * height, _ptr(Ka), _ptr(da),
                                                      0 if da is None else da.size, float(marker_size), int(bool(y_perp)), _ptr(out), self.cap, _ptr(n), C.byref(t)))
        return {"ticket": t.value, "ptrs": pa, "counts": ca, "out": out, "n": n, "K": Ka, "dist": da, "kind": "streams"}

    def wait(self, job):
        """arucohip_mgpu_wait: per-frame marker arrays of the job (frame order for a batch, [slot][frame] for streams)."""
        self._chk(self.L.arucohip_mgpu_wait(self.m, int(job["ticket"])))
        out, n = job["out"], job["n"]
        if job["kind"] == "batch":
            return [out[f, :n[f]].copy() for f in range(len(n))]
        ca = job["counts"]
        return [[out[g * self.per + j, :n[g * self.per + j]].copy() for j in range(int(ca[g]))] for g in range(self.G)]

    def detect_streams(self, ptrs, counts, width, height, K=None, dist=None, marker_size=-1.0, y_perp=False):
        """ptrs[g] = device pointer of slot g's frames (resident on its device), counts[g] frames each."""
        pa = (C.c_void_p * self.G)(*[C.c_void_p(int(x)) for x in ptrs])
        ca = np.ascontiguousarray(counts, dtype=np.int32)
        Ka, da = _f32(K), _f32(dist)
        out = np.zeros((self.G * self.per, self.cap), MARKER_DTYPE)
        n = np.zeros(self.G * self.per, np.int32)
        self._chk(self.L.arucohip_mgpu_detect_streams(self.m, pa, _ptr(ca), width, height, width, width * height, _ptr(Ka), _ptr(da),
                                                      0 if da is None else da.size, float(marker_size), int(bool(y_perp)), _ptr(out), self.cap, _ptr(n)))
        return [[out[g * self.per + j, :n[g * self.per + j]].copy() for j in range(int(ca[g]))] for g in range(self.G)]


# ---- OpenGL / Ogre conversions (host arithmetic)
def gl_modelview(rvec, tvec):
    """GetGLModelViewMatrix (src/utils.cpp:32-69): 16 doubles, column-major."""
    L = load()
    r, t, m = np.ascontiguousarray(rvec, np.float64), np.ascontiguousarray(tvec, np.float64), np.zeros(16, np.float64)
    rc = L.arucohip_gl_modelview(_ptr(r), _ptr(t), _ptr(m))
    if rc:
        raise ArucoHipError(rc, "gl_modelview")
    return m


def ogre_pose(rvec, tvec):
    """GetOgrePoseParameters (src/utils.cpp:71-147): (position[3], quaternion w,x,y,z)."""
    L = load()
    r, t = np.ascontiguousarray(rvec, np.float64), np.ascontiguousarray(tvec, np.float64)
    pos, q = np.zeros(3, np.float64), np.zeros(4, np.float64)
    rc = L.arucohip_ogre_pose(_ptr(r), _ptr(t), _ptr(pos), _ptr(q))
    if rc:
        raise ArucoHipError(rc, "ogre_pose")
    return pos, q


def gl_projection(K, cam_size, size, gnear, gfar, invert=False, ogre=False):
    """CameraParameters::glGetProjectionMatrix / OgreGetProjectionMatrix (src/cameraparameters.cpp:226-295)."""
    L = load()
    Ka, m = np.ascontiguousarray(K, np.float32).reshape(-1), np.zeros(16, np.float64)
    fn = L.arucohip_ogre_projection if ogre else L.arucohip_gl_projection
    rc = fn(_ptr(Ka), int(cam_size[0]), int(cam_size[1]), int(size[0]), int(size[1]), float(gnear), float(gfar), int(bool(invert)), _ptr(m))
    if rc:
        raise ArucoHipError(rc, "gl_projection")
    return m
