"""GPU parity tests of the SURVEY §8 rows built in round 2: optional erosion (n1), frame undistortion (f3), the locked-corner
pre-pass (a13). Integer / byte results bit-exact against the oracle, floating point within the tolerance written at the
assert."""
import numpy as np
import pytest

from tests.util import load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch  # noqa: F401
    from aruco_amd import capi, synth
    from oracle import orc

    assert torch.cuda.is_available()
    capi.load()
    return {"capi": capi, "orc": orc, "synth": synth, "torch": torch}


def erode3x3(img):
    """cv::erode(img, out, Mat()): 3x3 minimum, pixels outside the image never lower the result."""
    p = np.pad(img, 1, constant_values=255)
    out = np.full_like(img, 255)
    for dy in range(3):
        for dx in range(3):
            out = np.minimum(out, p[dy:dy + img.shape[0], dx:dx + img.shape[1]])
    return out


def test_erosion_flag(env):
    """north_star's optional erosion (row n1): the thresholded image handed out is the 3x3 erosion of the plain one and the
    contour stage works on it (borders = the sequential scan of the eroded image)."""
    capi, orc = env["capi"], env["orc"]
    fr, _ = env["synth"].make_stream(1, width=1920, height=1080, seed=3, device="cuda")
    for name, g in (("single", load_case("single")[0]), ("synth1080", fr[0].cpu().numpy())):
        hgt, wid = g.shape
        h = capi.Handle(wid, hgt, max_batch=1)
        try:
            plain = h.detect(g)
            thr = h.thresholded(0, g.shape)
            assert np.array_equal(thr, orc.adaptive_threshold(g, 7, 7.0))
            p = h.get_params()
            p.erode = 1
            h.set_params(p)
            eroded = h.detect(g)
            thr_e = h.thresholded(0, g.shape)
            assert np.array_equal(thr_e, erode3x3(thr)), name
            lo, hi = int(np.float32(0.04) * np.float32(max(wid, hgt)) * np.float32(4)), int(np.float32(0.5) * np.float32(max(wid, hgt)) * np.float32(4))
            ref = [c for c in orc.find_contours(thr_e) if lo < len(c["pts"]) < hi]
            got = h.debug_contours(0)
            assert len(got) == len(ref)
            for a, b in zip(got, ref):
                assert a["hole"] == b["hole"] and np.array_equal(a["pts"], b["pts"])
            # the 3-px threshold bands of the markers thin to 1 px and break at corners: erosion loses markers (which is why
            # the reference dropped the option); whatever is still found is a subset of the plain result
            assert set(int(m["id"]) for m in eroded) <= set(int(m["id"]) for m in plain)
            p.erode = 0
            h.set_params(p)
            assert h.detect(g).tobytes() == plain.tobytes()
        finally:
            h.close()


def test_erosion_on_tiles_equals_erosion_on_bytes(env, monkeypatch):
    """Round 3: with the byte image left as bit tiles + border lines (the default) the erosion runs on the tiles (64 pixels per 64-bit
    AND); with ARUCOHIP_THRES_BYTES=1 it runs on the bytes as before. Both equal numpy's 3x3 minimum, also where the image is not a whole
    number of tiles and where the frame pixels (kept in the border lines) are set."""
    capi, orc = env["capi"], env["orc"]
    fr, _ = env["synth"].make_stream(1, width=1920, height=1080, seed=5, device="cuda")
    full = fr[0].cpu().numpy()
    rng = np.random.default_rng(11)
    noisy = np.clip(full[:757, :1008].astype(np.int32) + rng.integers(-40, 40, (757, 1008)), 0, 255).astype(np.uint8)   # dark specks on the frame too
    for g in (full, np.ascontiguousarray(full[:757, :1008]), noisy, np.ascontiguousarray(full[100:612, 200:840])):
        hgt, wid = g.shape
        want = erode3x3(orc.adaptive_threshold(g, 7, 7.0))
        results = []
        for env_bytes in ("0", "1"):
            monkeypatch.setenv("ARUCOHIP_THRES_BYTES", env_bytes)
            h = capi.Handle(wid, hgt, max_batch=1)
            try:
                p = h.get_params()
                p.erode = 1
                h.set_params(p)
                m = h.detect(g)
                assert np.array_equal(h.thresholded(0, g.shape), want), (g.shape, env_bytes)
                results.append((m.tobytes(), [(c["hole"], c["pts"].tobytes()) for c in h.debug_contours(0)]))
            finally:
                h.close()
        assert results[0] == results[1], g.shape


def test_frame_undistort_bit_exact(env):
    """Row f3: cv::undistort on the device (map kernel + remap kernel) equals the CPU restatement byte for byte — gray and
    3-channel frames, 4 / 5 / 8 distortion coefficients, a batch, and a frame whose map leaves the image on every side; then
    the GL apps' sequence: undistort on the device, detect the device-resident result with an empty distortion vector."""
    capi, orc, torch = env["capi"], env["orc"], env["torch"]
    g, doc = load_case("single")
    K = np.array(doc["intrinsics"]["K"], np.float32)
    d5 = np.array(doc["intrinsics"]["dist"], np.float32)
    h = capi.Handle(1920, 1080, max_batch=2)
    try:
        for dist in (d5, d5[:4], np.array([-0.35, 0.2, 2e-3, -1e-3, -0.05, 0.01, -0.02, 0.003], np.float32), np.array([0.4, 0.3, 0, 0], np.float32)):
            got = h.undistort(g, K, dist)
            assert np.array_equal(got, orc.undistort(g, K, dist)), dist
        assert np.array_equal(h.undistort(g, K, None), orc.undistort(g, K, None))
        bgr = np.stack([g, (g // 2 + 17).astype(np.uint8), 255 - g], -1)
        assert np.array_equal(h.undistort(bgr, K, d5), orc.undistort(bgr, K, d5))
        fr, _ = env["synth"].make_stream(2, width=1920, height=1080, seed=9, device="cuda")
        K2 = np.array([1400, 0, 960, 0, 1400, 540, 0, 0, 1], np.float32)
        dd = np.array([-0.10, 0.02, 1e-3, -5e-4, 0], np.float32)
        frh = fr.cpu().numpy()
        got = h.undistort(frh, K2, dd)
        for f in range(2):
            assert np.array_equal(got[f], orc.undistort(frh[f], K2, dd))
        # device to device, then detect without distortion (aruco_test_gl.cpp:237-240)
        und = torch.empty_like(fr)
        Ka, da = np.ascontiguousarray(K2), np.ascontiguousarray(dd)
        import ctypes as C
        L = capi.load()
        rc = L.arucohip_undistort(h.h, C.c_void_p(fr.data_ptr()), 2, 1920, 1080, 1920, 1920 * 1080, 1, 1, Ka.ctypes.data_as(C.c_void_p),
                                  da.ctypes.data_as(C.c_void_p), 5, C.c_void_p(und.data_ptr()), 1)
        assert rc == 0
        out = torch.zeros((2, 64 * 96), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
        h.detect_batch_device(und.data_ptr(), 2, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr(), K=K2, marker_size=0.05)
        h.batch_status()
        assert np.array_equal(und.cpu().numpy(), got)
        arr = np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(2, 64)
        o = orc.Oracle()
        for f in range(2):
            ref = o.detect(got[f], K=K2, marker_size=0.05)
            n = int(cnt[f])
            assert [int(m["id"]) for m in arr[f, :n]] == [m["id"] for m in ref] and n >= 15
    finally:
        h.close()


def test_locked_corners(env):
    """Row a13: enableLockedCornersMethod (reference src/markerdetector.cpp:291-295 -> findCornerMaxima :157-199 before
    SUBPIX / HARRIS). The pre-pass yields integer positions: the device's Harris response equals the restatement's and the
    markers after the following refinement agree within the north_star corner tolerance (1e-4 relative)."""
    capi, orc = env["capi"], env["orc"]
    fr, _ = env["synth"].make_stream(1, width=1920, height=1080, seed=17, device="cuda")
    cases = [("single", load_case("single")[0]), ("board", load_case("board")[0]), ("synth1080", fr[0].cpu().numpy())]
    for method, mname in ((capi.CORNER_SUBPIX, "SUBPIX"), (capi.CORNER_HARRIS, "HARRIS"), (capi.CORNER_NONE, "NONE")):
        for name, g in cases:
            hgt, wid = g.shape
            h = capi.Handle(wid, hgt, max_batch=1)
            try:
                p = h.get_params()
                p.use_locked_corners, p.corner_method = 1, method
                h.set_params(p)
                got = h.detect(g)
                o = orc.Oracle(use_locked_corners=1, corner_method=method)
                ref = o.detect(g)
                assert [int(m["id"]) for m in got] == [m["id"] for m in ref], (mname, name)
                assert len(got) >= 5
                for a, b in zip(got, ref):
                    ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float).reshape(4, 2)
                    assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < 1e-4, (mname, name)
                if method != capi.CORNER_NONE:
                    # the pre-pass moves corners: the result differs from the plain method's
                    p.use_locked_corners = 0
                    h.set_params(p)
                    plain = h.detect(g)
                    moved = max(np.max(np.abs(np.asarray(a["corners"]) - np.asarray(b["corners"]))) for a, b in zip(got, plain) if a["id"] == b["id"])
                    assert moved > 0.01
            finally:
                h.close()


def test_canny_threshold_method(env):
    """Row a2's CANNY alternative (reference src/markerdetector.cpp:667-676: cv::Canny(grey, out, 10, 220)): the edge image of
    the device (suppression tiles + block-wise hysteresis to the fixed point) equals the restatement byte for byte on
    fixtures, noise, a frame that is not a multiple of 8 and a 1080p frame; detection with the method equals the oracle's."""
    capi, orc = env["capi"], env["orc"]
    rng = np.random.RandomState(21)
    fr, _ = env["synth"].make_stream(1, width=1920, height=1080, seed=5, device="cuda")
    base = ndimage_like_blur(rng.rand(203, 331))
    cases = [("single", load_case("single")[0]), ("board", load_case("board")[0]), ("smooth_331x203", (base * 255).astype(np.uint8)),
             ("noise_64x48", rng.randint(0, 256, (48, 64)).astype(np.uint8)), ("synth1080", fr[0].cpu().numpy())]
    h = capi.Handle(1920, 1080, max_batch=1)
    try:
        for name, g in cases:
            got = h.threshold(g, capi.THRES_CANNY)
            exp = orc.canny(g)
            assert np.array_equal(got, exp), (name, int((got != exp).sum()))
            assert 0 < exp.mean() < 128
        p = h.get_params()
        p.thres_method = capi.THRES_CANNY
        h.set_params(p)
        for name in ("single", "board"):
            g = load_case(name)[0]
            got = h.detect(g)
            ref = orc.Oracle(thres_method=2).detect(g)
            assert [int(m["id"]) for m in got] == [m["id"] for m in ref] and len(got) >= 5, name
            for a, b in zip(got, ref):
                ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float).reshape(4, 2)
                assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < 1e-4
            assert np.array_equal(h.thresholded(0, g.shape), orc.canny(g))
    finally:
        h.close()


def ndimage_like_blur(a):
    for _ in range(6):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5
    return (a - a.min()) / (a.max() - a.min())


def test_thresholded_image_kept_as_tiles_equals_the_byte_image(env, monkeypatch):
    """getThresholdedImage (src/markerdetector.h:183) after a batch: by default the 16-pixel-per-lane threshold kernel keeps the image as
    bit tiles + its four border lines and arucohip_get_thresholded expands the plane asked for; ARUCOHIP_THRES_BYTES=1 writes the bytes in
    the hot path. Both equal cv::adaptiveThreshold's restatement on every frame, borders included, also with a threshold range."""
    capi, orc, torch = env["capi"], env["orc"], env["torch"]
    from aruco_amd import synth
    fr, _ = synth.make_stream(4, seed=77, device="cuda")
    torch.cuda.synchronize()
    frames = fr.cpu().numpy()
    frames[1, 0, :] = 0; frames[1, -1, :] = 255; frames[1, :, 0] = 0; frames[1, :, -1] = 255     # something on the border lines
    frames[2, 0, ::2] = 0; frames[2, :, -1][::3] = 0
    for rng_ in (0, 1):
        p = capi.default_params()
        p.thres_param1_range = rng_
        got = {}
        for eager in ("0", "1"):
            monkeypatch.setenv("ARUCOHIP_THRES_BYTES", eager)
            h = capi.Handle(1920, 1080, max_batch=4, params=p)
            try:
                markers = h.detect_batch_host(frames)
                got[eager] = ([m.tobytes() for m in markers], [h.thresholded(f, (1080, 1920)) for f in range(4)])
            finally:
                h.close()
        assert got["0"][0] == got["1"][0]
        for f in range(4):
            assert np.array_equal(got["0"][1][f], got["1"][1][f]), (rng_, f)
            ref = orc.adaptive_threshold(frames[f], 7, 7.0)       # the middle plane of the range is the configured block size
            assert np.array_equal(got["0"][1][f], ref), (rng_, f)
