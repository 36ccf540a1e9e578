"""GPU tests of the limits and failure paths: what the device lists and packed fields cannot hold is reported through
the status code (never silently truncated, never out of bounds) and the shim path recovers the way the reference —
which has no list limits — behaves."""
import ctypes as C

import numpy as np
import pytest

from tests.util import load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch  # noqa: F401
    from aruco_amd import capi
    from oracle import orc

    assert torch.cuda.is_available()
    capi.load()
    return {"capi": capi, "orc": orc, "torch": torch}


def test_frame_wider_or_taller_than_the_handle_is_rejected(env):
    """the geometry check is per dimension (tile rows, 14-bit checkpoint coordinates), not per area."""
    capi = env["capi"]
    h = capi.Handle(1920, 1080, max_batch=1)
    try:
        for (w, hh) in ((3840, 270), (32, 64800 // 4), (1921, 1080), (1920, 1081)):
            g = np.zeros((hh, w), np.uint8)
            with pytest.raises(capi.ArucoHipError) as e:
                h.detect(g)
            assert e.value.code == capi.E_INVALID, (w, hh)
        # the same area in the handle's own shape is fine
        assert len(h.detect(np.full((1080, 1920), 128, np.uint8))) == 0
    finally:
        h.close()
    # handles whose raster index would not fit Quad::key are refused at creation
    with pytest.raises(capi.ArucoHipError):
        capi.Handle(16383, 16383, max_batch=1)


def test_board_with_too_many_points_is_refused(env):
    """arucohip_board_detect keeps obj + img + reprojection (7 floats per point) in an 8192-float scratch."""
    capi = env["capi"]
    h = capi.Handle(640, 480, max_batch=1)
    try:
        n = 300                                             # 1200 points: 5 * npts fits, 7 * npts does not
        markers = np.zeros(n, capi.MARKER_DTYPE)
        ids = np.arange(n, dtype=np.int32)
        obj = np.zeros((n, 4, 3), np.float32)
        for i in range(n):
            x0, y0 = (i % 20) * 30.0, (i // 20) * 30.0
            obj[i] = [[x0, y0, 0], [x0 + 20, y0, 0], [x0 + 20, y0 + 20, 0], [x0, y0 + 20, 0]]
            markers[i]["id"] = i
            # not an exact image of the plane: every point keeps a reprojection error far above 1e-9 px
            markers[i]["corners"] = (obj[i, :, :2] + 10 + np.random.RandomState(i).uniform(-0.4, 0.4, (4, 2))).reshape(-1)
        K = [500, 0, 320, 0, 500, 240, 0, 0, 1]
        with pytest.raises(capi.ArucoHipError) as e:
            h.board_detect(markers, ids, obj, capi.BOARD_PIX, K=K, dist=[0, 0, 0, 0], marker_size=0.05, repj_err_thres=1.5)
        assert e.value.code == capi.E_CAPACITY
        # a board that fits, with a reprojection threshold nothing passes: no pose, no crash (the reference would throw)
        res = h.board_detect(markers[:8], ids[:8], obj[:8], capi.BOARD_PIX, K=K, dist=[0, 0, 0, 0], marker_size=0.05, repj_err_thres=1e-9)
        assert res["has_pose"] == 0 and len(res["markers"]) == 8
    finally:
        h.close()


def _cluttered(rng, hgt, wid):
    """Dense texture as a GRAY frame: smooth random blobs of two levels with fine noise: thousands of long borders."""
    a = rng.randn(hgt // 12 + 2, wid // 12 + 2)
    a = np.kron(a, np.ones((12, 12)))[:hgt, :wid]
    for _ in range(3):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5
    g = np.where(a > 0, 200, 60) + rng.randint(-3, 4, size=a.shape)
    return np.clip(g, 0, 255).astype(np.uint8)


def test_cluttered_frame_overflow_is_reported_not_silent(env):
    """A cluttered 1080p frame on a handle with deliberately small lists: the out_on_device path reports the overflow
    through arucohip_batch_status; with default lists the same frame equals the oracle."""
    capi, orc, torch = env["capi"], env["orc"], env["torch"]
    rng = np.random.RandomState(5)
    g = _cluttered(rng, 1080, 1920)
    lim = capi.Limits()
    capi.load().arucohip_default_limits(C.byref(lim), 1920, 1080, 1)
    lim.long_walks_per_plane = 64
    lim.contours_per_frame = 64
    small = capi.Handle(1920, 1080, max_batch=1, limits=lim)
    try:
        fr = torch.from_numpy(g).cuda()
        out = torch.zeros((1, 64 * 96), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        small.detect_batch_device(fr.data_ptr(), 1, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
        with pytest.raises(capi.ArucoHipError) as e:
            small.batch_status()
        assert e.value.code == capi.E_OVERFLOW
        with pytest.raises(capi.ArucoHipError) as e:       # host path: same condition as the return code
            small.detect(g)
        assert e.value.code == capi.E_OVERFLOW
    finally:
        small.close()
    h = capi.Handle(1920, 1080, max_batch=1)
    try:
        h.detect(g)
        ref = orc.find_contours(orc.adaptive_threshold(g, 7, 7.0))
        ref = [c for c in ref if 307 < len(c["pts"]) < 3840]
        got = h.debug_contours(0)
        assert len(ref) > 100                              # it is cluttered
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert a["hole"] == b["hole"] and np.array_equal(a["pts"], b["pts"])
    finally:
        h.close()


def test_one_cluttered_frame_does_not_void_its_batch(env):
    """Per-frame overflow (round 3): a 32-frame batch of the synthetic stream with ONE cluttered frame in it, on a handle whose lists are
    deliberately small. The call reports ARUCOHIP_E_OVERFLOW, the cluttered frame comes back with n = -1 and the other 31 frames carry the
    bytes a generously sized handle returns; arucohip_detect_batch_retry_overflowed then redoes exactly that frame (host and device
    result arrays) and it equals the large handle's result too."""
    capi, torch = env["capi"], env["torch"]
    from aruco_amd import synth
    frames, _ = synth.make_stream(32, width=1920, height=1080, seed=77, device="cuda")
    fr = frames.cpu().numpy().copy()
    fr[13] = _cluttered(np.random.RandomState(5), 1080, 1920)
    big = capi.Handle(1920, 1080, max_batch=32)
    try:
        ref = big.detect_batch_host(fr, cap=64)
    finally:
        big.close()
    lim = capi.Limits()
    capi.load().arucohip_default_limits(C.byref(lim), 1920, 1080, 32)
    lim.long_walks_per_plane = 512          # the stream's frames need ~250 per plane and kind, the cluttered one thousands
    lim.contours_per_frame = 256
    small = capi.Handle(1920, 1080, max_batch=32, limits=lim)
    try:
        got, retried, first = small.detect_batch_host_tolerant(fr, cap=64, retry=False)
        assert list(np.nonzero(first < 0)[0]) == [13] and got[13] is None
        for f in range(32):
            if f != 13:
                assert got[f].tobytes() == ref[f].tobytes(), f
        got, retried, _ = small.detect_batch_host_tolerant(fr, cap=64)
        assert retried == 1
        for f in range(32):
            assert got[f].tobytes() == ref[f].tobytes(), f
        # device frames, device results
        dfr = torch.from_numpy(fr).cuda()
        out = torch.zeros((32, 64 * 96), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(32, dtype=torch.int32, device="cuda")
        small.detect_batch_device(dfr.data_ptr(), 32, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
        with pytest.raises(capi.ArucoHipError) as e:
            small.batch_status()
        assert e.value.code == capi.E_OVERFLOW
        assert int(cnt[13].item()) == -1 and int((cnt < 0).sum().item()) == 1
        assert small.retry_overflowed_device(dfr.data_ptr(), 32, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr()) == 1
        c = cnt.cpu().numpy()
        a = np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(32, 64)
        for f in range(32):
            assert a[f, :c[f]].tobytes() == ref[f].tobytes(), f
    finally:
        small.close()


def test_retry_handle_follows_decoder_changes(env):
    """The cached one-frame handle of arucohip_detect_batch_retry_overflowed must not outlive the decoder it was built with: retry (built-in
    decoder) -> install a caller's decoder that rejects everything -> retry again: the retried frame now has no markers, like every other
    frame of the batch; remove the callback -> the markers are back. (Round 3 kept the first retry handle: its frames were decoded with
    the old decoder.)"""
    capi, torch = env["capi"], env["torch"]
    from aruco_amd import synth
    frames, _ = synth.make_stream(4, width=1920, height=1080, seed=79, device="cuda")
    clut, _ = synth.make_stream(1, width=1920, height=1080, seed=5, device="cuda", clutter=True)
    fr = frames.cpu().numpy().copy()
    fr[2] = clut[0].cpu().numpy()
    big = capi.Handle(1920, 1080, max_batch=4)
    try:
        ref = big.detect_batch_host(fr, cap=64)
    finally:
        big.close()
    assert len(ref[2]) >= 10
    lim = capi.Limits()
    capi.load().arucohip_default_limits(C.byref(lim), 1920, 1080, 4)
    lim.long_walks_per_plane = 64
    lim.contours_per_frame = 128            # the textured frame keeps more borders than that, the flat ones about a hundred
    small = capi.Handle(1920, 1080, max_batch=4, limits=lim)
    try:
        got, retried, first = small.detect_batch_host_tolerant(fr, cap=64)
        assert first[2] < 0 and retried >= 1
        for f in range(4):
            assert got[f] is not None and (f != 2 or got[f].tobytes() == ref[f].tobytes())
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_int))
        fn = FN(lambda user, patch, size, nrot: -1)
        L = capi.load()
        assert L.arucohip_set_decoder_callback(small.h, C.cast(fn, C.c_void_p), None) == 0
        p = small.get_params()
        p.decoder_kind = 2
        small.set_params(p)
        got, retried, first = small.detect_batch_host_tolerant(fr, cap=64)
        assert first[2] < 0 and retried >= 1
        assert all(g is not None and len(g) == 0 for g in got)
        assert L.arucohip_set_decoder_callback(small.h, None, None) == 0
        got, retried, _ = small.detect_batch_host_tolerant(fr, cap=64)
        assert got[2].tobytes() == ref[2].tobytes()
    finally:
        small.close()
