"""GPU parity tests: the HIP path (through the C ABI, aruco_amd/libarucohip.so) against the CPU oracle on the same
inputs, and against the reference's golden vectors. Bit-exact for byte / integer / index results (threshold image,
contour point sequences, quads, ids, rotations, marker order); sub-pixel corners and poses within 1e-4 relative
(north_star tolerance) — tolerances are written next to each assert.
"""
import numpy as np
import pytest

from tests.util import load_case, rel_err

pytestmark = pytest.mark.gpu

CORNER_REL_TOL = 1e-4   # relative to the coordinate magnitude (north_star: "1e-4 relative")
POSE_REL_TOL = 1e-4


@pytest.fixture(scope="module")
def env():
    import torch  # noqa: F401  (load torch's HIP runtime first, see aruco_amd/capi.py)
    from aruco_amd import capi, synth
    from oracle import orc

    assert torch.cuda.is_available()
    capi.load()
    return {"capi": capi, "orc": orc, "synth": synth, "torch": torch}


@pytest.fixture(scope="module")
def handle(env):
    h = env["capi"].Handle(1920, 1080, max_batch=4)
    yield h
    h.close()


def blob_image(rng, h, w, scale=6, thr=0.0):
    """Random smooth blobs -> binary 0/255 image with many nested borders."""
    a = rng.randn(h // scale + 2, w // scale + 2)
    a = np.kron(a, np.ones((scale, scale)))[:h, :w]
    # cheap smoothing
    for _ in range(2):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5
    return ((a > thr) * 255).astype(np.uint8)


def gray_cases(env):
    rng = np.random.RandomState(7)
    cases = [(n, load_case(n)[0]) for n in ("single", "board", "chessboard")]
    cases.append(("noise_641x479", rng.randint(0, 256, size=(479, 641)).astype(np.uint8)))
    cases.append(("ramp_100x70", (np.add.outer(np.arange(70) * 3, np.arange(100) * 2) % 256).astype(np.uint8)))
    fr, _ = env["synth"].make_stream(1, width=1920, height=1080, seed=11, device="cuda")
    cases.append(("synth1080", fr[0].cpu().numpy()))
    return cases


def test_threshold_bit_exact(env, handle):
    capi, orc = env["capi"], env["orc"]
    for name, g in gray_cases(env):
        for block, c in ((7, 7.0), (3, 2.0), (21, 7.0), (9, -3.5)):
            got = handle.threshold(g, capi.THRES_ADPT, block, c)
            exp = orc.adaptive_threshold(g, block, c)
            assert np.array_equal(got, exp), (name, block, c, int((got != exp).sum()))
        got = handle.threshold(g, capi.THRES_FIXED, 100.0, 0.0)
        assert np.array_equal(got, np.where(g > 100, 0, 255).astype(np.uint8)), name


def _contour_check(env, handle, gray_or_bin, binary, min_size, max_size=0.5):
    capi, orc = env["capi"], env["orc"]
    p = handle.get_params()
    saved = (p.min_size, p.max_size)
    p.min_size, p.max_size = min_size, max_size
    handle.set_params(p)
    try:
        hgt, wid = gray_or_bin.shape
        if binary:
            handle.detect_rectangles(gray_or_bin)
            ref = orc.find_contours(gray_or_bin)
        else:
            handle.detect(gray_or_bin)
            ref = orc.find_contours(orc.adaptive_threshold(gray_or_bin, 7, 7.0))
        lo = int(np.float32(min_size) * np.float32(max(wid, hgt)) * np.float32(4))
        hi = int(np.float32(max_size) * np.float32(max(wid, hgt)) * np.float32(4))
        ref = [c for c in ref if lo < len(c["pts"]) < hi]
        got = handle.debug_contours(0)
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert a["hole"] == b["hole"]
            assert np.array_equal(a["pts"], b["pts"])   # same start pixel, same direction, every point
        return len(ref)
    finally:
        p.min_size, p.max_size = saved
        handle.set_params(p)


def test_contours_exact_on_blobs(env, handle):
    """Border following equals cv::findContours' scan on dense random blob images (nested holes, thin bridges)."""
    rng = np.random.RandomState(3)
    total = 0
    for (h, w, scale) in ((240, 320, 6), (479, 641, 4), (300, 300, 3), (200, 520, 10), (128, 128, 2)):
        img = blob_image(rng, h, w, scale)
        total += _contour_check(env, handle, img, True, 0.004, 1.0)
    # pure salt-and-pepper: single pixels, diagonal chains, 1-px lines
    img = ((rng.rand(160, 200) > 0.55) * 255).astype(np.uint8)
    total += _contour_check(env, handle, img, True, 0.002, 1.0)
    assert total > 200


def test_contours_exact_on_frames(env, handle):
    for name, g in gray_cases(env):
        if min(g.shape) < 100:
            continue
        _contour_check(env, handle, g, False, 0.01)


def _compare_markers(got, exp, corner_tol=CORNER_REL_TOL, pose=False):
    assert [int(m["id"]) for m in got] == [m["id"] for m in exp]
    for a, b in zip(got, exp):
        ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float).reshape(4, 2)
        assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < corner_tol
        if pose:
            assert int(a["has_pose"]) == 1
            assert rel_err(a["rvec"], b["rvec"]) < POSE_REL_TOL
            assert rel_err(a["tvec"], b["tvec"]) < POSE_REL_TOL


def test_candidates_and_decode_exact(env, handle):
    orc = env["orc"]
    for name, g in gray_cases(env):
        if min(g.shape) < 100:
            continue
        o = orc.Oracle()
        o.detect(g)
        handle.detect(g)
        q, ids, nrot = handle.debug_candidates(0)
        ref = o.candidates()
        assert len(q) == len(ref), name
        for i, r in enumerate(ref):
            assert np.array_equal(q[i], r["quad0"]), (name, i)     # integer quads, reference order
            assert ids[i] == r["id"], (name, i)
            if r["id"] >= 0:
                assert nrot[i] == r["nrot"], (name, i)
        rej = handle.candidates(0)
        assert len(rej) == len(o.rejected())
        assert np.array_equal(handle.thresholded(0, g.shape), o.thresholded())


def test_warp_bit_exact(env, handle):
    orc = env["orc"]
    g, _ = load_case("board")
    o = orc.Oracle()
    o.detect(g)
    for c in o.candidates():
        for size in (56, 28):
            assert np.array_equal(handle.warp(g, c["quad0"], size), orc.warp(g, c["quad0"], size))


def test_golden_single(env, handle):
    """Reference test Aruco.Single (test/core_tests.cpp:77-116) through the HIP path."""
    g, doc = load_case("single")
    intr = doc["intrinsics"]
    got = handle.detect(g, K=intr["K"], dist=intr["dist"], marker_size=1.0)
    exp = [{"id": e["id"], "corners": e["corners"], "rvec": e["Rvec"], "tvec": e["Tvec"]} for e in doc["markers"]]
    _compare_markers(got, exp, pose=True)
    ref = env["orc"].Oracle().detect(g, K=intr["K"], dist=intr["dist"], marker_size=1.0)
    _compare_markers(got, ref, corner_tol=1e-6, pose=True)


@pytest.mark.parametrize("name", ["board", "chessboard"])
def test_golden_boards(env, handle, name):
    """Reference tests Aruco.Board / Aruco.Multi (test/core_tests.cpp:164-228) through the HIP path."""
    capi, orc = env["capi"], env["orc"]
    g, doc = load_case(name)
    intr, bc = doc["intrinsics"], doc["board_conf"]
    got = handle.detect(g)
    b = handle.board_detect(got, bc["ids"], bc["obj"], bc["info_type"], intr["K"], intr["dist"], 1.0)
    _compare_markers(b["markers"], doc["markers"])
    assert rel_err(b["rvec"], doc["board"]["Rvec"]) < POSE_REL_TOL
    assert rel_err(b["tvec"], doc["board"]["Tvec"]) < POSE_REL_TOL
    assert abs(b["prob"] - len(doc["markers"]) / len(bc["ids"])) < 1e-6
    ob = orc.board_detect(orc.Oracle().detect(g), bc["ids"], bc["obj"], bc["info_type"], intr["K"], intr["dist"], 1.0)
    assert rel_err(b["rvec"], ob["rvec"]) < 1e-6 and rel_err(b["tvec"], ob["tvec"]) < 1e-6
    # reprojection-error filter + y-perpendicular variant (aruco_test_board configuration)
    b2 = handle.board_detect(got, bc["ids"], bc["obj"], bc["info_type"], intr["K"], intr["dist"], 1.0, repj_err_thres=1.5, y_perp=True)
    o2 = orc.board_detect(orc.Oracle().detect(g), bc["ids"], bc["obj"], bc["info_type"], intr["K"], intr["dist"], 1.0, 1.5, True)
    assert rel_err(b2["rvec"], o2["rvec"]) < POSE_REL_TOL and rel_err(b2["tvec"], o2["tvec"]) < POSE_REL_TOL


@pytest.mark.parametrize("method", ["NONE", "HARRIS", "SUBPIX", "LINES"])
def test_corner_methods_vs_oracle(env, handle, method):
    capi, orc = env["capi"], env["orc"]
    code = {"NONE": 0, "HARRIS": 1, "SUBPIX": 2, "LINES": 3}[method]
    p = handle.get_params()
    saved = p.corner_method
    p.corner_method = code
    handle.set_params(p)
    try:
        for name in ("single", "board"):
            g, doc = load_case(name)
            intr = doc["intrinsics"]
            got = handle.detect(g, K=intr["K"], dist=intr["dist"], marker_size=0.05, y_perp=(name == "board"))
            ref = orc.Oracle(corner_method=code).detect(g, K=intr["K"], dist=intr["dist"], marker_size=0.05, y_perp=(name == "board"))
            _compare_markers(got, ref, pose=True)
    finally:
        p.corner_method = saved
        handle.set_params(p)


def test_synthetic_1080p_batch(env, handle):
    """Config 2/3 frames: batch == per-frame == oracle; detected ids are the rendered ids, corners within 1 px of truth."""
    orc, synth = env["orc"], env["synth"]
    fr, truth = synth.make_stream(3, seed=4711, device="cuda")
    frames = fr.cpu().numpy()
    K = [1400, 0, 960, 0, 1400, 540, 0, 0, 1]
    dist = [-0.10, 0.02, 1e-3, -5e-4, 0]
    batch = handle.detect_batch_host(frames, K=K, dist=dist, marker_size=0.05)
    o = orc.Oracle()
    for f in range(len(frames)):
        ref = o.detect(frames[f], K=K, dist=dist, marker_size=0.05)
        _compare_markers(batch[f], ref, pose=True)
        single = handle.detect(frames[f], K=K, dist=dist, marker_size=0.05)
        assert single.tobytes() == batch[f].tobytes()
        tq = {t["id"]: t["quad"] for t in truth[f]}
        assert set(int(m["id"]) for m in batch[f]) <= set(tq)
        assert len(batch[f]) >= len(tq) - 1
        for m in batch[f]:
            c = np.asarray(m["corners"], float).reshape(4, 2)
            q = tq[int(m["id"])]
            assert min(np.max(np.linalg.norm(np.roll(q, k, axis=0) - c, axis=1)) for k in range(4)) < 1.5


def test_device_resident_batch(env, handle):
    """Frames and results stay in HBM (torch tensors by raw pointer), the path bench.py times."""
    torch, synth, capi = env["torch"], env["synth"], env["capi"]
    fr, _ = synth.make_stream(2, seed=5, device="cuda")
    cap = 64
    out = torch.zeros((2, cap * 96), dtype=torch.uint8, device="cuda")
    n = torch.zeros(2, dtype=torch.int32, device="cuda")
    handle.detect_batch_device(fr.data_ptr(), 2, 1920, 1080, out.data_ptr(), cap, n.data_ptr())
    handle.batch_status()
    host = handle.detect_batch_host(fr.cpu().numpy())
    nn = n.cpu().numpy()
    arr = np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(2, cap)
    for f in range(2):
        assert nn[f] == len(host[f])
        assert arr[f, :nn[f]].tobytes() == host[f].tobytes()


def test_error_codes(env, handle):
    capi = env["capi"]
    p = handle.get_params()
    p.min_size, p.max_size = 0.6, 0.5
    with pytest.raises(capi.ArucoHipError) as e:
        handle.set_params(p)
    assert e.value.code == capi.E_INVALID
    p = handle.get_params()
    p.warp_size = 5
    with pytest.raises(capi.ArucoHipError):
        handle.set_params(p)
    p = handle.get_params()
    p.warp_size = 200                       # a value the device kernels are not built for (patches up to 128x128)
    with pytest.raises(capi.ArucoHipError) as e:
        handle.set_params(p)
    assert e.value.code == capi.E_UNSUPPORTED
    p = handle.get_params()
    p.thres_method = 7                      # not a ThresholdMethods value
    with pytest.raises(capi.ArucoHipError) as e:
        handle.set_params(p)
    assert e.value.code == capi.E_INVALID
    g, _ = load_case("board")
    with pytest.raises(capi.ArucoHipError) as e:
        handle.detect(g, cap=3)
    assert e.value.code == capi.E_CAPACITY
    with pytest.raises(capi.ArucoHipError) as e:
        handle.board_detect(handle.detect(g), [], np.zeros((0, 4, 3)), 0, None, None, 1.0)
    assert e.value.code == capi.E_BOARD_CONFIG


def test_multi_threshold_range(env):
    """setThresholdParamRange (markerdetector.cpp:322-333): 2r+1 threshold planes per frame, candidates joined in plane
    order — exact vs the oracle (row f2 of SURVEY §8f)."""
    capi, orc = env["capi"], env["orc"]
    for rng_ in (1, 2):
        p = capi.default_params()
        p.thres_param1_range = rng_
        h = capi.Handle(640, 480, max_batch=2, params=p)
        try:
            for name in ("single", "board"):
                g, doc = load_case(name)
                intr = doc["intrinsics"]
                o = orc.Oracle(thres_range=rng_)
                ref = o.detect(g, K=intr["K"], dist=intr["dist"], marker_size=0.05)
                got = h.detect(g, K=intr["K"], dist=intr["dist"], marker_size=0.05)
                _compare_markers(got, ref, pose=True)
                q, ids, nrot = h.debug_candidates(0)
                refc = o.candidates()
                assert len(q) == len(refc)
                for i, r in enumerate(refc):
                    assert np.array_equal(q[i], r["quad0"]) and ids[i] == r["id"]
                assert np.array_equal(h.thresholded(0, g.shape), o.thresholded())   # the middle plane
        finally:
            h.close()


def test_4k_board_frames(env):
    """Config 4 of BASELINE.json: the 6x4 board of testdata/board/board_pix.yml rendered at 3840x2160 through random
    poses; MarkerDetector + BoardDetector on device vs the oracle."""
    capi, orc, synth = env["capi"], env["orc"], env["synth"]
    _, doc = load_case("board")
    bc = doc["board_conf"]
    W, H = 3840, 2160
    sx, sy = W / 640.0, H / 480.0                      # CameraParameters::resize rule (cameraparameters.cpp:173-178)
    K0 = np.array(doc["intrinsics"]["K"], np.float32).reshape(3, 3)
    K = K0.copy()
    K[0, 0] *= np.float32(sx); K[0, 2] *= np.float32(sx); K[1, 1] *= np.float32(sy); K[1, 2] *= np.float32(sy)
    dist = [0.0, 0.0, 0.0, 0.0, 0.0]
    rng = np.random.RandomState(9)
    h = capi.Handle(W, H, max_batch=2)
    try:
        frames = []
        for f in range(2):
            rvec = rng.uniform(-0.2, 0.2, 3)   # board_pix.yml: x right, y down, z away from the camera
            tvec = np.array([rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), rng.uniform(0.52, 0.60)])
            fr, quads = synth.render_board(bc["ids"], bc["obj"], K.astype(float), rvec, tvec, W, H, rng, device="cuda",
                                           unit=0.039 / 100.0)
            frames.append(fr.cpu().numpy())
        got = h.detect_batch_host(np.stack(frames))
        batched = h.board_detect_batch(2, bc["ids"], bc["obj"], bc["info_type"], K.reshape(-1), dist, 0.039)
        batched_f = h.board_detect_batch(2, bc["ids"], bc["obj"], bc["info_type"], K.reshape(-1), dist, 0.039, repj_err_thres=4.0, y_perp=True)
        o = orc.Oracle()
        for f in range(2):
            ref = o.detect(frames[f])
            assert len(ref) >= 20 and set(m["id"] for m in ref) <= set(bc["ids"])
            _compare_markers(got[f], ref)
            b = h.board_detect(got[f], bc["ids"], bc["obj"], bc["info_type"], K.reshape(-1), dist, 0.039)
            ob = orc.board_detect(ref, bc["ids"], bc["obj"], bc["info_type"], K.reshape(-1), dist, 0.039)
            assert b["has_pose"] == 1 and abs(b["prob"] - ob["prob"]) < 1e-6
            assert rel_err(b["rvec"], ob["rvec"]) < POSE_REL_TOL and rel_err(b["tvec"], ob["tvec"]) < POSE_REL_TOL
            # batched wave-parallel board pose == per-frame call == oracle
            bb = batched[f]
            assert bb["has_pose"] == 1 and bb["n_markers"] == len(ob["markers"]) and abs(bb["prob"] - ob["prob"]) < 1e-6
            assert rel_err(bb["rvec"], ob["rvec"]) < POSE_REL_TOL and rel_err(bb["tvec"], ob["tvec"]) < POSE_REL_TOL
            of = orc.board_detect(ref, bc["ids"], bc["obj"], bc["info_type"], K.reshape(-1), dist, 0.039, 4.0, True)
            assert batched_f[f]["has_pose"] == of["has_pose"] == 1
            assert rel_err(batched_f[f]["rvec"], of["rvec"]) < POSE_REL_TOL and rel_err(batched_f[f]["tvec"], of["tvec"]) < POSE_REL_TOL
    finally:
        h.close()


@pytest.mark.gpu
def test_bgr_input(env, handle):
    """BGR frames are converted on the device exactly like cv::cvtColor(BGR2GRAY) (oracle restatement),
    aligned and ragged widths; detection on a BGR frame equals detection on its gray conversion."""
    orc = env["orc"]
    rng = np.random.RandomState(21)
    for (h, w) in ((64, 96), (75, 131), (480, 640)):
        bgr = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        assert np.array_equal(handle.bgr_to_gray(bgr), orc.bgr2gray(bgr))
    g, doc = load_case("single")
    intr = doc["intrinsics"]
    # a colour image whose gray conversion is NOT the plain channel: tint the channels, keep the result in range
    bgr = np.stack([np.clip(g.astype(int) + 9, 0, 255), g.astype(int), np.clip(g.astype(int) - 7, 0, 255)], axis=2).astype(np.uint8)
    gray = orc.bgr2gray(bgr)
    a = handle.detect_bgr(bgr, K=intr["K"], dist=intr["dist"], marker_size=1.0)
    b = handle.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
    assert len(a) > 0 and a.tobytes() == b.tobytes()
    batch = handle.detect_batch_bgr_host(np.stack([bgr, bgr[:, ::-1].copy()]), K=intr["K"], dist=intr["dist"], marker_size=1.0)
    assert batch[0].tobytes() == a.tobytes()


def test_segment_pipeline_mode(env, monkeypatch):
    """ARUCOHIP_CONTOURS=segments (fully parallel waypoint-segment border extraction): same borders, same markers."""
    capi = env["capi"]
    monkeypatch.setenv("ARUCOHIP_CONTOURS", "segments")
    h = capi.Handle(1920, 1080, max_batch=2)
    monkeypatch.delenv("ARUCOHIP_CONTOURS")
    ref = capi.Handle(1920, 1080, max_batch=2)
    try:
        rng = np.random.RandomState(7)
        total = _contour_check(env, h, blob_image(rng, 240, 320, 6), True, 0.004, 1.0)
        total += _contour_check(env, h, ((rng.rand(120, 160) > 0.55) * 255).astype(np.uint8), True, 0.002, 1.0)
        assert total > 50
        for name in ("single", "board"):
            g, doc = load_case(name)
            intr = doc["intrinsics"]
            a = h.detect(g, K=intr["K"], dist=intr["dist"], marker_size=1.0)
            b = ref.detect(g, K=intr["K"], dist=intr["dist"], marker_size=1.0)
            assert len(a) > 0 and a.tobytes() == b.tobytes()
    finally:
        h.close()
        ref.close()


@pytest.mark.parametrize("grid", ["8", "16", "32"])
def test_segment_pipeline_grids_on_1080p_frames(env, monkeypatch, grid):
    """The waypoint-segment pipeline (the default for one frame per call since round 4 up to 1080p; round 4 rebuilt its step: the cracks of a visit as
    bit sets, the run rule from the lane's block) at the grid spacings 8 / 16 / 32: every kept border of a cluttered and a flat 1080p frame, of dense
    blobs and of salt-and-pepper noise equals cv::findContours' (start pixel, direction, every point), with the reference's size filter and a permissive one."""
    capi, synth = env["capi"], env["synth"]
    monkeypatch.setenv("ARUCOHIP_CONTOURS", "segments")
    monkeypatch.setenv("ARUCOHIP_GRID", grid)
    h = capi.Handle(1920, 1080, max_batch=1)
    monkeypatch.delenv("ARUCOHIP_CONTOURS")
    monkeypatch.delenv("ARUCOHIP_GRID")
    try:
        fc, _ = synth.make_stream(1, seed=31, device="cpu", clutter=True)
        ff, _ = synth.make_stream(1, seed=32, device="cpu")
        total = 0
        for g in (fc[0].numpy(), ff[0].numpy()):
            total += _contour_check(env, h, g, False, 0.04, 0.5)
            total += _contour_check(env, h, g, False, 0.02, 0.5)      # (a permissive filter on these frames outgrows the candidate lists)
        rng = np.random.RandomState(11)
        total += _contour_check(env, h, blob_image(rng, 479, 641, 4), True, 0.004, 1.0)
        total += _contour_check(env, h, ((rng.rand(200, 300) > 0.5) * 255).astype(np.uint8), True, 0.002, 1.0)
        assert total > 600
    finally:
        h.close()


def test_chunk_streams(env, monkeypatch):
    """ARUCOHIP_STREAMS=3: a batch cut into chunks on forked streams gives the bytes of the single-stream batch, and the
    per-frame getters / batched board pose find their frame's worker."""
    capi, synth = env["capi"], env["synth"]
    fr, _ = synth.make_stream(7, seed=99, device="cuda")
    frames = fr.cpu().numpy()
    K = [1400, 0, 960, 0, 1400, 540, 0, 0, 1]
    monkeypatch.setenv("ARUCOHIP_STREAMS", "3")
    h = capi.Handle(1920, 1080, max_batch=7)
    monkeypatch.delenv("ARUCOHIP_STREAMS")
    ref = capi.Handle(1920, 1080, max_batch=7)
    try:
        a = h.detect_batch_host(frames, K=K, dist=[0.0] * 5, marker_size=0.05)
        b = ref.detect_batch_host(frames, K=K, dist=[0.0] * 5, marker_size=0.05)
        assert h.batch_chunks() == (3, 3) and ref.batch_chunks()[0] == 1
        for f in range(len(frames)):
            assert len(a[f]) > 0 and a[f].tobytes() == b[f].tobytes()
        for f in (0, 3, 6):   # one frame of every chunk
            assert np.array_equal(h.thresholded(f, frames[f].shape), ref.thresholded(f, frames[f].shape))
            ca, cb = h.debug_contours(f), ref.debug_contours(f)
            assert len(ca) == len(cb) and all(np.array_equal(x["pts"], y["pts"]) for x, y in zip(ca, cb))
        assert h.batch_status() == 0
    finally:
        h.close()
        ref.close()


def test_hrm_decoder(env):
    """highly reliable markers (reference test Aruco.HRM_Single, test/core_tests.cpp:310-353) through
    the HIP path: dictionary d4x4_100, the test's detector settings; equals the CPU restatement and the reference's golden."""
    capi, orc = env["capi"], env["orc"]
    gray, doc = load_case("hrm")
    intr, st, dic = doc["intrinsics"], doc["settings"], doc["dictionary"]
    h = capi.Handle(640, 640, max_batch=2)   # also takes the rotated 480x640 frame below (limits are per dimension)
    try:
        p = h.get_params()
        p.thres_param1, p.thres_param2, p.min_size, p.max_size, p.warp_size = st["thres_param1"], st["thres_param2"], st["min_size"], st["max_size"], st["warp_size"]
        h.set_params(p)
        h.set_dictionary(dic["markers"], dic["tau0"])
        got = h.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=st["marker_size"])
        o = orc.Oracle(thres_p1=st["thres_param1"], thres_p2=st["thres_param2"], min_size=st["min_size"], max_size=st["max_size"],
                       warp_size=st["warp_size"])
        o.set_hrm_dictionary(dic["markers"], dic["tau0"])
        ref = o.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=st["marker_size"])
        _compare_markers(got, ref, pose=True)
        assert [int(m["id"]) for m in got] == [e["id"] for e in doc["markers"]]
        for m, e in zip(got, doc["markers"]):
            assert rel_err(m["rvec"], e["Rvec"]) < POSE_REL_TOL and rel_err(m["tvec"], e["Tvec"]) < POSE_REL_TOL
        # every candidate's decode result (id and rotation), not only the accepted markers
        dq, did, drot = h.debug_candidates(0)
        cands = o.candidates()
        assert [int(v) for v in did] == [c["id"] for c in cands]
        assert all(int(a) == c["nrot"] for a, c in zip(drot, cands) if c["id"] >= 0)
        # a rotated frame exercises the other rotations; a corrupted dictionary entry exercises the error correction
        got90 = h.detect(np.ascontiguousarray(np.rot90(gray)), marker_size=-1.0)
        ref90 = o.detect(np.ascontiguousarray(np.rot90(gray)))
        _compare_markers(got90, ref90)
        bad = list(dic["markers"])
        bad[3] = ("0" if bad[3][0] == "1" else "1") + bad[3][1:]          # one wrong bit: still within (tau0-1)/2 = 1
        h.set_dictionary(bad, dic["tau0"])
        o.set_hrm_dictionary(bad, dic["tau0"])
        _compare_markers(h.detect(gray), o.detect(gray))
        assert 3 in [int(m["id"]) for m in h.detect(gray)]
        h.set_dictionary(None, 0)                                           # back to the fiducial decoder
        o.set_hrm_dictionary(None, 0)
        _compare_markers(h.detect(gray), o.detect(gray))
    finally:
        h.close()


def test_hrm_larger_dictionaries(env):
    """The reference's own 5x5 ... 8x8 dictionaries (testdata/hrm/dictionaries/d5x5_100 ... d8x8_100.yml, committed as
    tests/golden/hrm_dictionaries.json; 8x8 = 64-bit codes, all 64 lanes vote): device == CPU restatement on synthetic frames."""
    from tests.util import load_hrm_dictionary
    capi, orc, synth = env["capi"], env["orc"], env["synth"]
    for n in (5, 6, 7, 8):
        D, tau = load_hrm_dictionary(n)
        fr, lay = synth.make_hrm_frame(D, width=1280, height=720, seed=7 + n, n_markers=10, device="cuda")
        g = fr.cpu().numpy()
        h = capi.Handle(1280, 720, max_batch=1)
        try:
            p = h.get_params()
            p.warp_size = (n + 2) * 8
            h.set_params(p)
            h.set_dictionary(D, tau)
            o = orc.Oracle(warp_size=(n + 2) * 8)
            o.set_hrm_dictionary(D, tau)
            got, ref = h.detect(g), o.detect(g)
            _compare_markers(got, ref)
            assert [int(m["id"]) for m in got] == sorted(m["id"] for m in lay)
            _, did, drot = h.debug_candidates(0)
            cands = o.candidates()
            assert [int(v) for v in did] == [c["id"] for c in cands]
        finally:
            h.close()


def test_refine_fail_frame(env):
    """Reference test Aruco.RefineFail (test/core_tests.cpp:355-382) through the HIP path: goes through and equals the
    CPU restatement (degenerate LINES fits included)."""
    import os
    from tests.util import read_pgm, GOLDEN
    capi, orc = env["capi"], env["orc"]
    gray = read_pgm(os.path.join(GOLDEN, "hrm_refine_fail.pgm"))
    _, doc = load_case("hrm")
    intr, dic = doc["intrinsics"], doc["dictionary"]
    h = capi.Handle(640, 480, max_batch=1)
    try:
        p = h.get_params()
        p.thres_param1, p.thres_param2, p.min_size, p.max_size, p.warp_size = 21, 7, 0.005, 0.5, 48
        h.set_params(p)
        h.set_dictionary(dic["markers"], dic["tau0"])
        o = orc.Oracle(thres_p1=21, thres_p2=7, min_size=0.005, max_size=0.5, warp_size=48)
        o.set_hrm_dictionary(dic["markers"], dic["tau0"])
        got = h.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
        ref = o.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
        _compare_markers(got, ref, pose=True)
    finally:
        h.close()


def test_threshold_wide_kernel_strip_edges(env):
    """The 16-pixel-per-lane threshold kernel (widths that are multiples of 16, blocks up to 9x9): widths around the 1024-pixel
    strip (one strip exactly, one strip + one lane, two strips, a partly filled second strip), heights that are not multiples
    of the 128-row segment or of 8, every block size of that kernel, positive and negative C; threshold bytes and contours
    (which go through the tiles the kernel writes) equal the oracle."""
    capi, orc = env["capi"], env["orc"]
    rng = np.random.RandomState(11)
    h = capi.Handle(2064, 300, max_batch=1)
    try:
        for (wid, hgt) in ((1024, 64), (1040, 131), (2048, 40), (2064, 137), (16, 300), (1920, 33), (48, 32)):
            base = rng.randint(0, 256, size=(hgt // 8 + 2, wid // 8 + 2)).astype(np.float32)
            g = np.kron(base, np.ones((8, 8), np.float32))[:hgt, :wid] + rng.randint(-20, 21, size=(hgt, wid))
            g = np.clip(g, 0, 255).astype(np.uint8)
            if wid >= 32 and hgt >= 32:
                # 7x7 blocks take the round-3 kernel (test constants folded into the column sums: every sign and size of C matters; |C| > 200
                # falls back to the round-2 kernel), the other block sizes the round-2 one
                for block, c in ((7, 7.0), (3, 2.0), (5, -3.0), (9, 11.5), (7, -30.5), (7, -3.0), (7, 0.0), (7, 0.5), (7, 13.9), (7, 100.0),
                                 (7, 200.0), (7, 201.0), (7, -200.0), (7, 250.0)):
                    got = h.threshold(g, capi.THRES_ADPT, block, c)
                    exp = orc.adaptive_threshold(g, block, c)
                    assert np.array_equal(got, exp), (wid, hgt, block, c, int((got != exp).sum()))
        # extremes of the sums: all-black, all-white and a hard checkerboard, with the constants that push the folded offsets both ways
        for fill in ("black", "white", "checker"):
            g = np.zeros((96, 1040), np.uint8) if fill == "black" else np.full((96, 1040), 255, np.uint8)
            if fill == "checker":
                g = (((np.arange(96)[:, None] // 3 + np.arange(1040)[None, :] // 5) % 2) * 255).astype(np.uint8)
            for c in (7.0, -7.0, 0.0, 199.0, -199.0):
                got = h.threshold(g, capi.THRES_ADPT, 7, c)
                exp = orc.adaptive_threshold(g, 7, c)
                assert np.array_equal(got, exp), (fill, c, int((got != exp).sum()))
        for (wid, hgt) in ((1040, 131), (2064, 137)):
            img = blob_image(rng, hgt, wid, 5)
            g = np.where(img > 0, 40, 200).astype(np.uint8)       # gray frame whose threshold bands are the blob borders
            _contour_check(env, h, g, False, 0.01, 1.0)
    finally:
        h.close()
