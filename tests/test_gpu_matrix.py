"""Parameter matrix on synthetic 1080p batches: every combination below runs a 6-frame batch through the HIP path and the
oracle with the same MarkerDetector settings (threshold block / constant / range, FIXED threshold, corner method, warp size,
size filter, border distance) and with camera intrinsics + lens distortion, so that the branches of the reference that the
640x480 goldens do not reach (other block sizes, SUBPIX / HARRIS with distortion, other patch sizes) are compared at the
bench's frame size. ids / order exact, corners and poses <= 1e-4 relative."""
import numpy as np
import pytest

from tests.util import rel_err

pytestmark = pytest.mark.gpu

K = [1400, 0, 960, 0, 1400, 540, 0, 0, 1]
DIST = [-0.10, 0.02, 1e-3, -5e-4, 0]
NF = 6

# (thres_method, p1, p2, range, corner_method, warp_size, min_size, max_size, border_dist)
CASES = [
    (1, 7, 7, 0, 3, 56, 0.04, 0.5, 0.025),     # defaults
    (1, 5, 6, 0, 3, 56, 0.04, 0.5, 0.025),     # 5x5 block
    (1, 9, 9, 0, 2, 56, 0.04, 0.5, 0.025),     # 9x9 block, SUBPIX (window = 9)
    (1, 11, 7, 0, 1, 56, 0.04, 0.5, 0.025),    # 11x11 block (narrow threshold kernel), HARRIS
    (1, 13, 8, 0, 0, 56, 0.04, 0.5, 0.025),    # 13x13 block (32-bit sums), corners as detected
    (1, 7, 7, 1, 3, 56, 0.04, 0.5, 0.025),     # threshold range 1: three planes per frame
    (0, 128, 0, 0, 3, 56, 0.04, 0.5, 0.025),   # FIXED threshold
    (1, 7, 7, 0, 3, 28, 0.04, 0.5, 0.025),     # 28x28 patches
    (1, 7, 7, 0, 3, 63, 0.04, 0.5, 0.025),     # 63x63 patches (not a multiple of 8)
    (1, 7, 7, 0, 3, 56, 0.06, 0.2, 0.05),      # tighter size filter, wider border band
    (1, 7, 7, 0, 2, 56, 0.02, 0.9, 0.0),       # loose size filter, no border band, SUBPIX
]


@pytest.fixture(scope="module")
def env():
    import torch
    from aruco_amd import capi, synth
    from oracle import orc

    assert torch.cuda.is_available()
    capi.load()
    fr, truth = synth.make_stream(NF, seed=977, device="cuda")
    torch.cuda.synchronize()
    return {"capi": capi, "orc": orc, "frames": fr.cpu().numpy(), "truth": truth}


@pytest.mark.parametrize("case", CASES, ids=lambda c: "m%d_b%d_c%d_r%d_corner%d_ws%d_min%g_max%g_bd%g" % c)
def test_parameter_matrix_on_1080p_batches(env, case):
    capi, orc = env["capi"], env["orc"]
    tm, p1, p2, rng_, cm, ws, mn, mx, bd = case
    p = capi.default_params()
    p.thres_method, p.thres_param1, p.thres_param2, p.thres_param1_range = tm, p1, p2, rng_
    p.corner_method, p.warp_size, p.min_size, p.max_size, p.border_dist = cm, ws, mn, mx, bd
    h = capi.Handle(1920, 1080, max_batch=NF, params=p)
    try:
        got = h.detect_batch_host(env["frames"], K=K, dist=DIST, marker_size=0.05)
    finally:
        h.close()
    o = orc.Oracle(thres_method=tm, thres_p1=p1, thres_p2=p2, thres_range=rng_, corner_method=cm, warp_size=ws, min_size=mn, max_size=mx,
                   border_dist=bd)
    found = 0
    for f in range(NF):
        ref = o.detect(env["frames"][f], K=K, dist=DIST, marker_size=0.05)
        assert [int(m["id"]) for m in got[f]] == [m["id"] for m in ref], f
        for a, b in zip(got[f], ref):
            ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float).reshape(4, 2)
            assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < 1e-4, f          # 1e-4 relative (north_star)
            assert int(a["has_pose"]) == 1
            assert rel_err(a["rvec"], b["rvec"]) < 1e-4 and rel_err(a["tvec"], b["tvec"]) < 1e-4, f
        found += len(ref)
    if tm == 1 and mx >= 0.5:
        assert found >= 15 * NF          # the adaptive settings find nearly every rendered marker


def test_zero_distortion_vector_with_poses_equals_the_oracle(env):
    """Config 3's first form (intrinsics with dist = zeros(5)): a NON-EMPTY distortion vector of zeros takes the undistort / distort
    branches of LINES (markerdetector.cpp:956-959, 989-991) and of solvePnP with a model that changes nothing - compared with the oracle WITH
    poses on the 1080p frames (round 3 compared this case only between two HIP handles)."""
    capi, orc = env["capi"], env["orc"]
    zeros = [0.0] * 5
    h = capi.Handle(1920, 1080, max_batch=NF)
    try:
        got = h.detect_batch_host(env["frames"], K=K, dist=zeros, marker_size=0.05)
        nodist = h.detect_batch_host(env["frames"], K=K, dist=None, marker_size=0.05)
    finally:
        h.close()
    o = orc.Oracle()
    found = 0
    for f in range(NF):
        ref = o.detect(env["frames"][f], K=K, dist=zeros, marker_size=0.05)
        assert [int(m["id"]) for m in got[f]] == [m["id"] for m in ref] == [int(m["id"]) for m in nodist[f]], f
        for a, b, c in zip(got[f], ref, nodist[f]):
            ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float).reshape(4, 2)
            assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < 1e-4, f
            assert int(a["has_pose"]) == 1
            assert rel_err(a["rvec"], b["rvec"]) < 1e-4 and rel_err(a["tvec"], b["tvec"]) < 1e-4, f
            # and the zero model is (numerically) no model: the same pose as without a distortion vector
            assert rel_err(a["tvec"], c["tvec"]) < 1e-4, f
        found += len(ref)
    assert found >= 15 * NF


# (width, height, row stride, x offset, y offset) of a window of the 1080p frames, handed over with that row stride
GEOMETRIES = [
    (1283, 727, 1283, 300, 200),     # odd width and stride: byte-wise threshold kernel
    (1280, 720, 1280, 320, 180),     # 16-byte aligned rows: wide kernel, 2 strips (the second one partial)
    (1000, 1000, 1024, 400, 40),     # width a multiple of 8 only, padded stride
    (1920, 1080, 2048, 0, 0),        # full frame with padded rows
    (1024, 1024, 1024, 500, 30),     # exactly one 1-KiB strip
    (1040, 600, 1040, 100, 300),     # one strip + 16 pixels
    (644, 483, 644, 900, 400),       # multiple of 4 only: dword kernel
]


@pytest.mark.parametrize("geo", GEOMETRIES, ids=lambda g: "%dx%d_stride%d" % g[:3])
def test_frame_geometries(env, geo):
    """Frames of other sizes, alignments and row strides (windows of the synthetic 1080p frames, device-resident with the given
    row stride): thresholded image and every marker against the oracle on the same window."""
    import torch
    capi, orc = env["capi"], env["orc"]
    w, hgt, stride, x0, y0 = geo
    win = np.ascontiguousarray(env["frames"][:3, y0:y0 + hgt, x0:x0 + w])
    assert win.shape == (3, hgt, w)
    buf = np.zeros((3, hgt, stride), np.uint8)
    buf[:, :, :w] = win
    buf[:, :, w:] = 255 - (np.arange(stride - w, dtype=np.uint8) * 37)[None, None, :] if stride > w else 0   # padding must not matter
    dev = torch.from_numpy(buf).cuda()
    out = torch.zeros((3, 64 * 96), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(3, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    h = capi.Handle(w, hgt, max_batch=3)
    try:
        h.detect_batch_device(dev.data_ptr(), 3, w, hgt, out.data_ptr(), 64, cnt.data_ptr(), K=K, dist=DIST, marker_size=0.05, row_stride=stride,
                              frame_stride=stride * hgt)
        h.batch_status()
        torch.cuda.synchronize()
        arr = np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(3, 64)
        n = cnt.cpu().numpy()
        o = orc.Oracle()
        total = 0
        for f in range(3):
            ref = o.detect(win[f], K=K, dist=DIST, marker_size=0.05)
            assert np.array_equal(h.thresholded(f, (hgt, w)), o.thresholded()), f
            got = arr[f, :n[f]]
            assert [int(m["id"]) for m in got] == [m["id"] for m in ref], f
            for a, b in zip(got, ref):
                ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float).reshape(4, 2)
                assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < 1e-4, f
                assert rel_err(a["rvec"], b["rvec"]) < 1e-4 and rel_err(a["tvec"], b["tvec"]) < 1e-4, f
            total += len(ref)
        assert total > 0
    finally:
        h.close()


def test_4k_board_with_harris_and_reprojection_filter(env):
    """SURVEY §8d config 4, second form: the aruco_test_board configuration (utils/aruco_test_board.cpp:146-149) — HARRIS corner refinement and
    set_repj_err_thres(1.5) — on 3840x2160 board frames: markers and both board-pose entries (per frame, batched) against the oracle."""
    import torch
    from tests.util import load_case
    from aruco_amd import synth
    capi, orc = env["capi"], env["orc"]
    W, H, NB = 3840, 2160, 3
    _, doc = load_case("board")
    bc = doc["board_conf"]
    Kb = np.array(doc["intrinsics"]["K"], np.float32).reshape(3, 3)
    Kb[0, 0] *= np.float32(W / 640.0); Kb[0, 2] *= np.float32(W / 640.0)
    Kb[1, 1] *= np.float32(H / 480.0); Kb[1, 2] *= np.float32(H / 480.0)
    Kf = Kb.reshape(-1)
    dist = [0.0] * 5
    frames, _ = synth.make_board_stream(NB, bc["ids"], bc["obj"], Kf, width=W, height=H, seed=31, device="cuda")
    torch.cuda.synchronize()
    host = frames.cpu().numpy()
    p = capi.default_params()
    p.corner_method = capi.CORNER_HARRIS if hasattr(capi, "CORNER_HARRIS") else 1
    h = capi.Handle(W, H, max_batch=NB, params=p)
    try:
        got = h.detect_batch_host(host)
        batched = h.board_detect_batch(NB, bc["ids"], bc["obj"], bc["info_type"], Kf, dist, 0.039, repj_err_thres=1.5)
        o = orc.Oracle(corner_method=1)
        for f in range(NB):
            ref = o.detect(host[f])
            assert [int(m["id"]) for m in got[f]] == [m["id"] for m in ref] and len(ref) >= 20
            for a, b in zip(got[f], ref):
                ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float).reshape(4, 2)
                assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < 1e-4
            ob = orc.board_detect(ref, bc["ids"], bc["obj"], bc["info_type"], Kf, dist, 0.039, 1.5, False)
            one = h.board_detect(got[f], bc["ids"], bc["obj"], bc["info_type"], Kf, dist, 0.039, repj_err_thres=1.5)
            for bb in (one, batched[f]):
                assert bb["has_pose"] == ob["has_pose"] == 1
                assert rel_err(bb["rvec"], ob["rvec"]) < 1e-4 and rel_err(bb["tvec"], ob["tvec"]) < 1e-4
    finally:
        h.close()


@pytest.mark.parametrize("shape", [(96, 8192), (72, 16368), (4096, 48), (40, 4112)], ids=lambda s: "%dx%d" % (s[1], s[0]))
def test_extreme_aspect_frames(env, shape):
    """Very wide and very tall frames (up to the 14-bit coordinate limit; 8 and 16 strips of the non-empty-tile bitmap, whose words one
    wave-load fetches per tile row): thresholded image and every kept border against the oracle's sequential scan."""
    capi, orc = env["capi"], env["orc"]
    hgt, wid = shape
    rng = np.random.RandomState(hgt * 7 + wid)
    a = rng.randn(hgt // 6 + 2, wid // 6 + 2)
    a = np.kron(a, np.ones((6, 6)))[:hgt, :wid]
    for _ in range(2):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5
    g = np.clip(np.where(a > 0, 190, 70) + rng.randint(-3, 4, size=a.shape), 0, 255).astype(np.uint8)
    p = capi.default_params()
    p.min_size, p.max_size = 0.002, 0.9
    import ctypes as C
    lim = capi.Limits()
    capi.load().arucohip_default_limits(C.byref(lim), wid, hgt, 1)
    lim.points_per_frame *= 8                      # a dense texture keeps far more border points than a camera frame of this area
    lim.contours_per_frame *= 4
    h = capi.Handle(wid, hgt, max_batch=1, params=p, limits=lim)
    try:
        h.detect(g)
        thr = orc.adaptive_threshold(g, 7, 7.0)
        assert np.array_equal(h.thresholded(0, g.shape), thr)
        lo = int(np.float32(0.002) * np.float32(max(wid, hgt)) * np.float32(4))
        hi = int(np.float32(0.9) * np.float32(max(wid, hgt)) * np.float32(4))
        ref = [c for c in orc.find_contours(thr) if lo < len(c["pts"]) < hi]
        got = h.debug_contours(0)
        assert len(ref) > 20 and len(got) == len(ref)
        for x, y in zip(got, ref):
            assert x["hole"] == y["hole"] and np.array_equal(x["pts"], y["pts"])
    finally:
        h.close()


def test_degenerate_frames(env):
    """Constant and checkered frames, frames down to one pixel, and an empty batch: nothing faults, thresholded image, borders and markers
    equal the restatement."""
    import torch
    capi, orc = env["capi"], env["orc"]
    rng = np.random.RandomState(2)
    frames = [np.zeros((480, 640), np.uint8), np.full((480, 640), 255, np.uint8), (np.indices((480, 640)).sum(0) % 2 * 255).astype(np.uint8),
              np.zeros((32, 32), np.uint8), rng.randint(0, 256, (32, 32)).astype(np.uint8), rng.randint(0, 256, (33, 47)).astype(np.uint8),
              rng.randint(0, 256, (1040, 32)).astype(np.uint8), rng.randint(0, 256, (32, 1040)).astype(np.uint8)]
    import ctypes as C
    for i, g in enumerate(frames):
        hgt, wid = g.shape
        lim = capi.Limits()
        capi.load().arucohip_default_limits(C.byref(lim), wid, hgt, 2)
        if i == 2:
            # every second pixel of the checkerboard is a border of its own: 150 k start candidates. The default lists report that
            # (the shim doubles them and detects again); sized for it the frame goes through
            h = capi.Handle(wid, hgt, max_batch=2)
            try:
                with pytest.raises(capi.ArucoHipError) as e:
                    h.detect(g)
                assert e.value.code == capi.E_OVERFLOW
            finally:
                h.close()
            lim.triggers_per_frame = 400000
        h = capi.Handle(wid, hgt, max_batch=2, limits=lim)
        try:
            got = h.detect(g)
            ref = orc.Oracle().detect(g)
            assert [int(m["id"]) for m in got] == [m["id"] for m in ref]
            assert np.array_equal(h.thresholded(0, g.shape), orc.adaptive_threshold(g, 7, 7.0)), g.shape
            both = h.detect_batch_host(np.stack([g, g]))
            assert len(both) == 2 and all(len(b) == len(ref) for b in both)
        finally:
            h.close()
    h = capi.Handle(640, 480, max_batch=4)
    try:
        out = torch.zeros((1, 64 * 96), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        fr = torch.zeros((1, 480, 640), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        # zero frames: either refused as an invalid argument or a no-op; never a fault
        try:
            h.detect_batch_device(fr.data_ptr(), 0, 640, 480, out.data_ptr(), 64, cnt.data_ptr())
        except capi.ArucoHipError as e:
            assert e.code == capi.E_INVALID
        assert len(h.detect(np.full((480, 640), 128, np.uint8))) == 0        # the handle is still usable
    finally:
        h.close()
    # frames far smaller than the smallest handle (32 x 32), down to one pixel, in both border pipelines (a one-frame handle uses the
    # waypoint segments, a larger one the walkers): thresholded image, kept borders and markers against the oracle
    def blob(hgt, wid):
        a = rng.randn(hgt // 3 + 2, wid // 3 + 2)
        a = np.kron(a, np.ones((3, 3)))[:hgt, :wid]
        return np.clip(np.where(a > 0, 200, 60) + rng.randint(-3, 4, size=a.shape), 0, 255).astype(np.uint8)
    for mb in (1, 2):
        for (hgt, wid) in ((17, 9), (8, 8), (1, 64), (64, 1), (16, 33), (31, 31), (5, 300), (300, 5), (24, 24), (25, 40), (2, 2), (1, 1)):
            g = blob(hgt, wid)
            h = capi.Handle(max(wid, 32), max(hgt, 32), max_batch=mb)
            try:
                got = h.detect(g)
                thr = orc.adaptive_threshold(g, 7, 7.0)
                assert np.array_equal(h.thresholded(0, g.shape), thr), (mb, hgt, wid)
                lo = int(np.float32(0.04) * np.float32(max(wid, hgt)) * np.float32(4))
                hi = int(np.float32(0.5) * np.float32(max(wid, hgt)) * np.float32(4))
                if lo >= 1:        # the walkers never keep 1-point borders (no quad can come from one)
                    ref = [c for c in orc.find_contours(thr) if lo < len(c["pts"]) < hi]
                    gc = h.debug_contours(0)
                    assert len(gc) == len(ref) and all(x["hole"] == y["hole"] and np.array_equal(x["pts"], y["pts"]) for x, y in zip(gc, ref)), (mb, hgt, wid)
                assert [int(m["id"]) for m in got] == [m["id"] for m in orc.Oracle().detect(g)]
            finally:
                h.close()
    h = capi.Handle(640, 480, max_batch=4)
    try:
        assert len(h.detect(np.zeros((17, 9), np.uint8))) == 0
    finally:
        h.close()


def test_cluttered_stream_equals_the_oracle(env):
    """The robustness leg of bench.py (`--clutter`: a two-level blob texture behind the 20 markers, about twice the kept borders and
    contour points of the flat stream): 32 such 1080p frames through the HIP path as one batch equal the oracle frame by frame — ids,
    order and count exact, LINES corners <= 1e-4 relative — and every rendered marker is found."""
    import torch
    from aruco_amd import synth
    capi, orc = env["capi"], env["orc"]
    fr, truth = synth.make_stream(32, seed=4711, device="cuda", clutter=True)
    torch.cuda.synchronize()
    frames = fr.cpu().numpy()
    h = capi.Handle(1920, 1080, max_batch=32)
    try:
        got = h.detect_batch_host(frames, cap=64)
        fill = h.debug_counters()
    finally:
        h.close()
    assert fill["status"] == 0 and fill["contours"] / 32 > 120          # it is cluttered (the flat stream keeps ~95 borders per frame)
    o = orc.Oracle()
    total = 0
    for f in range(32):
        ref = o.detect(frames[f])
        assert [int(m["id"]) for m in got[f]] == [m["id"] for m in ref], f
        for a, b in zip(got[f], ref):
            ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float)
            assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < 1e-4, f
        assert set(t["id"] for t in truth[f]) == set(int(m["id"]) for m in got[f]), f
        total += len(got[f])
    assert total == 32 * 20


def test_two_borders_per_wave_equal_one_border_per_wave(env, monkeypatch):
    """Round 4: contour_quad takes two borders of at most 512 points per wave (one per half wave). On 16 cluttered and 16 flat 1080p frames the
    candidates of every frame (quads in reference order, ids, rotations), the emitted contour points and the markers are byte-identical to
    the one-border-per-wave path (ARUCOHIP_QUAD_DUAL=0), and the candidate quads equal the oracle's."""
    import torch
    from aruco_amd import synth
    capi, orc = env["capi"], env["orc"]
    fr_c, _ = synth.make_stream(16, seed=99, device="cuda", clutter=True)
    fr_f, _ = synth.make_stream(16, seed=98, device="cuda")
    torch.cuda.synchronize()
    frames = np.concatenate([fr_c.cpu().numpy(), fr_f.cpu().numpy()])
    res = {}
    for dual in ("1", "0"):
        monkeypatch.setenv("ARUCOHIP_QUAD_DUAL", dual)
        h = capi.Handle(1920, 1080, max_batch=32)
        try:
            got = h.detect_batch_host(frames, cap=64)
            assert h.debug_counters()["status"] == 0
            cands = [h.debug_candidates(f) for f in range(32)]
            conts = [h.debug_contours(f) for f in (0, 5, 17, 31)]
            res[dual] = (got, cands, conts)
        finally:
            h.close()
    monkeypatch.delenv("ARUCOHIP_QUAD_DUAL")
    a, b = res["1"], res["0"]
    for f in range(32):
        assert a[0][f].tobytes() == b[0][f].tobytes(), f
        for x, y in zip(a[1][f], b[1][f]):
            assert x.tobytes() == y.tobytes(), f
    for ca, cb in zip(a[2], b[2]):
        assert len(ca) == len(cb) > 50
        for x, y in zip(ca, cb):
            assert x["hole"] == y["hole"] and x["start"] == y["start"] and x["pts"].tobytes() == y["pts"].tobytes()
    o = orc.Oracle()
    for f in (0, 9, 16, 31):
        o.detect_raw(frames[f])
        ref = o.candidates()
        q = a[1][f][0]
        assert len(ref) == len(q), f
        for i, r in enumerate(ref):
            assert np.array_equal(q[i], r["quad0"]), (f, i)     # integer quads, reference order
            assert a[1][f][1][i] == r["id"], (f, i)


def test_otsu_thresholds_of_every_candidate_equal_the_oracle(env):
    """Round 4: otsu_kernel takes the denominator's half of the fp64 division (reciprocal of q1 + two Newton steps) off the chain of dependent
    operations. The threshold it leaves for EVERY candidate - markers and rejected quads alike - of 8 cluttered and 8 flat 1080p frames and of the
    reference's four stills equals getThreshVal_Otsu_8u's (oracle, order-dependent double recurrence) on the candidate's 56x56 patch."""
    import torch
    from aruco_amd import synth
    from tests.util import load_case
    capi, orc = env["capi"], env["orc"]
    fr_c, _ = synth.make_stream(8, seed=7, device="cuda", clutter=True)
    fr_f, _ = synth.make_stream(8, seed=8, device="cuda")
    torch.cuda.synchronize()
    frames = np.concatenate([fr_c.cpu().numpy(), fr_f.cpu().numpy()])
    checked = 0
    h = capi.Handle(1920, 1080, max_batch=16)
    try:
        h.detect_batch_host(frames, cap=64)
        o = orc.Oracle()
        for f in range(16):
            q, ids, _ = h.debug_candidates(f)
            thr = h.debug_otsu(f)
            o.detect_raw(frames[f])
            ref = o.candidates()
            assert len(ref) == len(q) == len(thr), f
            for i, r in enumerate(ref):
                assert np.array_equal(q[i], r["quad0"]), (f, i)
                assert thr[i] == orc.otsu(orc.warp(frames[f], r["quad0"], 56)), (f, i)
                checked += 1
    finally:
        h.close()
    for name in ("single", "board", "chessboard"):
        g, _ = load_case(name)
        h = capi.Handle(g.shape[1], g.shape[0], max_batch=1)
        try:
            h.detect(g)
            q, _, _ = h.debug_candidates(0)
            thr = h.debug_otsu(0)
            for i in range(len(q)):
                assert thr[i] == orc.otsu(orc.warp(g, q[i], 56)), (name, i)
                checked += 1
        finally:
            h.close()
    assert checked > 500
