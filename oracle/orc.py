"""ORACLE — TEST INFRASTRUCTURE ONLY. ctypes binding of oracle/liborc.so (the CPU restatement of the reference path).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg. Never from aruco_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborc.so")


class OrcMarker(C.Structure):
    _fields_ = [("id", C.c_int32), ("corners", C.c_float * 8), ("ssize", C.c_float), ("has_pose", C.c_int32),
                ("pad_", C.c_int32), ("rvec", C.c_double * 3), ("tvec", C.c_double * 3)]


class OrcParams(C.Structure):
    _fields_ = [("thres_method", C.c_int32), ("thres_p1", C.c_double), ("thres_p2", C.c_double),
                ("thres_range", C.c_int32), ("corner_method", C.c_int32), ("min_size", C.c_float),
                ("max_size", C.c_float), ("warp_size", C.c_int32), ("border_dist", C.c_float),
                ("use_locked_corners", C.c_int32), ("approx_inner_product", C.c_int32)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orc_imgproc.cpp", "orc_pnp.cpp", "orc_detect.cpp", "orc_capi.cpp", "orc.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liborc.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_find_contours.restype = C.c_void_p
        L.orc_board_detect.restype = C.c_float
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def _f32(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


def marker_dict(m):
    return {"id": int(m.id), "corners": np.array(m.corners, dtype=np.float32).reshape(4, 2).copy(),
            "ssize": float(m.ssize), "has_pose": int(m.has_pose), "rvec": np.array(m.rvec), "tvec": np.array(m.tvec)}


class Oracle:
    """Mirror of aruco::MarkerDetector on the CPU restatement."""

    def __init__(self, **params):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create())
        self.set_params(**params)

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def get_params(self):
        p = OrcParams()
        self.L.orc_get_params(self.h, C.byref(p))
        return p

    def set_params(self, **kw):
        p = self.get_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
        self.L.orc_set_params(self.h, C.byref(p))

    def set_hrm_dictionary(self, markers, tau0, rate=1.0):
        """HighlyReliableMarkers::loadDictionary + setMakerDetectorFunction(HighlyReliableMarkers::detect).
        markers: bit strings of n*n characters (row-major, '1' = white); None / empty restores the fiducial decoder."""
        codes = hrm_codes(markers or [])
        n = int(round(len(markers[0]) ** 0.5)) if markers else 0
        arr = (C.c_uint64 * max(len(codes), 1))(*codes)
        self.L.orc_set_hrm(self.h, n, len(codes), arr, int(tau0), C.c_float(rate))

    def detect(self, gray, K=None, dist=None, marker_size=-1.0, y_perp=False, cap=256):
        g, gp = _u8(gray)
        h, w = g.shape
        Ka, Kp = _f32(K)
        da, dp = _f32(dist)
        nd = 0 if da is None else da.size
        out = (OrcMarker * cap)()
        n = C.c_int(0)
        rc = self.L.orc_detect(self.h, gp, w, h, w, Kp, dp, nd, C.c_float(marker_size), int(bool(y_perp)), out, cap, C.byref(n))
        if rc != 0:
            raise RuntimeError("oracle detect rc=%d" % rc)
        self._shape = (h, w)
        return [marker_dict(out[i]) for i in range(min(n.value, cap))]

    def detect_raw(self, gray, K=None, dist=None, marker_size=-1.0, y_perp=False, cap=256):
        """Same as detect() but returns (ctypes array, n) — used by the timed cpu_baseline loop."""
        g, gp = _u8(gray)
        h, w = g.shape
        Ka, Kp = _f32(K)
        da, dp = _f32(dist)
        nd = 0 if da is None else da.size
        out = (OrcMarker * cap)()
        n = C.c_int(0)
        self.L.orc_detect(self.h, gp, w, h, w, Kp, dp, nd, C.c_float(marker_size), int(bool(y_perp)), out, cap, C.byref(n))
        return out, n.value

    def thresholded(self):
        h, w = self._shape
        out = np.empty((h, w), np.uint8)
        self.L.orc_get_thresholded(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def contours(self):
        res = []
        for i in range(self.L.orc_num_contours(self.h)):
            hole, tx, ty = C.c_int(), C.c_int(), C.c_int()
            n = self.L.orc_contour_info(self.h, i, C.byref(hole), C.byref(tx), C.byref(ty))
            pts = np.empty((n, 2), np.int32)
            self.L.orc_contour_points(self.h, i, pts.ctypes.data_as(C.c_void_p))
            res.append({"hole": hole.value, "trig": (tx.value, ty.value), "pts": pts})
        return res

    def candidates(self, with_contour=False):
        res = []
        for i in range(self.L.orc_num_candidates(self.h)):
            q0 = np.empty((4, 2), np.float32)
            q = np.empty((4, 2), np.float32)
            cid, nrot, idx, nc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            self.L.orc_candidate(self.h, i, q0.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), C.byref(cid),
                                 C.byref(nrot), C.byref(idx), C.byref(nc))
            d = {"quad0": q0, "quad": q, "id": cid.value, "nrot": nrot.value, "idx": idx.value, "ncontour": nc.value}
            if with_contour:
                pts = np.empty((nc.value, 2), np.int32)
                self.L.orc_candidate_contour(self.h, i, pts.ctypes.data_as(C.c_void_p))
                d["contour"] = pts
            res.append(d)
        return res

    def rejected(self):
        res = []
        for i in range(self.L.orc_num_rejected(self.h)):
            q = np.empty((4, 2), np.float32)
            self.L.orc_rejected(self.h, i, q.ctypes.data_as(C.c_void_p))
            res.append(q)
        return res


def adaptive_threshold(gray, block=7, c=7.0):
    g, gp = _u8(gray)
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_adaptive_threshold(gp, w, h, w, int(block), C.c_double(c), out.ctypes.data_as(C.c_void_p))
    return out


def find_contours(binimg):
    b, bp = _u8(binimg)
    h, w = b.shape
    L = lib()
    s = C.c_void_p(L.orc_find_contours(bp, w, h))
    res = []
    for i in range(L.orc_cset_size(s)):
        hole, tx, ty = C.c_int(), C.c_int(), C.c_int()
        n = L.orc_cset_info(s, i, C.byref(hole), C.byref(tx), C.byref(ty))
        pts = np.empty((n, 2), np.int32)
        L.orc_cset_points(s, i, pts.ctypes.data_as(C.c_void_p))
        res.append({"hole": hole.value, "trig": (tx.value, ty.value), "pts": pts})
    L.orc_cset_free(s)
    return res


def approx_poly(pts, eps, inner_product_rule=1, cap=4096):
    p = np.ascontiguousarray(pts, dtype=np.int32)
    out = np.empty((cap, 2), np.int32)
    n = lib().orc_approx_poly(p.ctypes.data_as(C.c_void_p), len(p), C.c_double(eps), inner_product_rule,
                              out.ctypes.data_as(C.c_void_p), cap)
    return out[:n].copy()


def warp(gray, quad, size=56):
    g, gp = _u8(gray)
    h, w = g.shape
    q, qp = _f32(quad)
    out = np.empty((size, size), np.uint8)
    lib().orc_warp(gp, w, h, w, qp, size, out.ctypes.data_as(C.c_void_p))
    return out


def fiducial_detect(patch):
    p, pp = _u8(patch)
    nrot = C.c_int(0)
    mid = lib().orc_fiducial_detect(pp, p.shape[0], C.byref(nrot))
    return mid, nrot.value


def hrm_codes(markers):
    """Dictionary bit strings -> uint64 codes, bit y*n+x = cell (y, x)."""
    return [sum(1 << i for i, ch in enumerate(m) if ch == "1") for m in markers]


def bgr2gray(bgr):
    """cv::cvtColor(BGR2GRAY) for 8-bit data (orc_imgproc.cpp; call site src/markerdetector.cpp:307-310)."""
    b = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w, c = b.shape
    assert c == 3
    g = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(b.ctypes.data_as(C.c_void_p), h * w, g.ctypes.data_as(C.c_void_p))
    return g


def solve_pnp(obj, img, K, dist):
    o, op = _f32(obj)
    m, mp = _f32(img)
    Ka, Kp = _f32(K)
    da, dp = _f32(dist)
    r = (C.c_double * 3)()
    t = (C.c_double * 3)()
    ok = lib().orc_solve_pnp(op, mp, len(o.reshape(-1, 3)), Kp, dp, 0 if da is None else da.size, r, t)
    return bool(ok), np.array(r), np.array(t)


def refine_lines(contour, corners, K=None, dist=None):
    """MarkerDetector::refineCandidateLines on a contour (n x 2 ints) and four corners; returns the refined corners (4 x 2)."""
    xy = np.ascontiguousarray(contour, dtype=np.int32).reshape(-1, 2)
    c = np.ascontiguousarray(corners, dtype=np.float32).reshape(8).copy()
    Ka, Kp = _f32(K)
    da, dp = _f32(dist)
    lib().orc_refine_lines(xy.ctypes.data_as(C.c_void_p), len(xy), c.ctypes.data_as(C.c_void_p), Kp, dp, 0 if da is None else da.size)
    return c.reshape(4, 2)


def corner_subpix(gray, pts, win=7, max_iter=8, eps=0.005):
    g, gp = _u8(gray)
    h, w = g.shape
    p = np.ascontiguousarray(pts, dtype=np.float32).copy()
    lib().orc_corner_subpix(gp, w, h, w, p.ctypes.data_as(C.c_void_p), len(p.reshape(-1, 2)), win, max_iter, C.c_double(eps))
    return p


def corner_harris(gray, pts):
    g, gp = _u8(gray)
    h, w = g.shape
    p = np.ascontiguousarray(pts, dtype=np.float32).copy()
    lib().orc_corner_harris(gp, w, h, w, p.ctypes.data_as(C.c_void_p), len(p.reshape(-1, 2)))
    return p


def undistort(img, K, dist):
    """cv::undistort(img, out, K, dist) for an 8-bit image [H][W] or [H][W][3]."""
    a = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    Ka, da = _f32(K)[0], _f32(dist)[0]
    out = np.empty_like(a)
    lib().orc_undistort(a.ctypes.data_as(C.c_void_p), w, h, w * cn, cn, Ka.ctypes.data_as(C.c_void_p),
                        None if da is None else da.ctypes.data_as(C.c_void_p), 0 if da is None else da.size, out.ctypes.data_as(C.c_void_p))
    return out


def canny(gray, low=10, high=220):
    """cv::Canny(gray, out, low, high) with the 3x3 Sobel and the L1 magnitude."""
    g, gp = _u8(gray)
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_canny(gp, w, h, w, int(low), int(high), out.ctypes.data_as(C.c_void_p))
    return out


def find_corner_maxima(gray, pts, wsize):
    g, gp = _u8(gray)
    h, w = g.shape
    p = np.ascontiguousarray(pts, dtype=np.float32).copy()
    lib().orc_find_corner_maxima(gp, w, h, w, p.ctypes.data_as(C.c_void_p), len(p.reshape(-1, 2)), int(wsize))
    return p


def corner_harris_window(gray, x0, y0, x1, y1):
    g, gp = _u8(gray)
    h, w = g.shape
    out = np.zeros((y1 - y0, x1 - x0), np.float32)
    lib().orc_corner_harris_window(gp, w, h, w, x0, y0, x1, y1, out.ctypes.data_as(C.c_void_p))
    return out


def board_detect(markers, ids, obj, info_type, K, dist, marker_size, repj_thres=-1.0, y_perp=False):
    n = len(markers)
    ms = (OrcMarker * max(n, 1))()
    for i, m in enumerate(markers):
        ms[i].id = m["id"]
        for k, v in enumerate(np.asarray(m["corners"], np.float32).reshape(-1)):
            ms[i].corners[k] = v
        ms[i].ssize = m.get("ssize", -1.0)
    ida = np.ascontiguousarray(ids, dtype=np.int32)
    oa, op = _f32(obj)
    Ka, Kp = _f32(K)
    da, dp = _f32(dist)
    out = (OrcMarker * max(n, 1))()
    nout, hp = C.c_int(0), C.c_int(0)
    r = (C.c_double * 3)()
    t = (C.c_double * 3)()
    prob = lib().orc_board_detect(ms, n, ida.ctypes.data_as(C.c_void_p), op, len(ida), info_type, Kp, dp,
                                  0 if da is None else da.size, C.c_float(marker_size), C.c_float(repj_thres),
                                  int(bool(y_perp)), out, C.byref(nout), r, t, C.byref(hp))
    return {"prob": float(prob), "markers": [marker_dict(out[i]) for i in range(nout.value)], "rvec": np.array(r),
            "tvec": np.array(t), "has_pose": hp.value}


def otsu(img):
    a, ap = _u8(img)
    return int(lib().orc_otsu(ap, a.size))


def rotate_x_axis(rvec):
    r = (C.c_double * 3)(*[float(v) for v in rvec])
    lib().orc_rotate_x_axis(r)
    return np.array(r)
