"""MarkerDetector::refineCandidateLines as a stage entry point (reference src/markerdetector.h:280, .cpp:931-997; SURVEY.md §8b "must stay
callable"): arucohip_refine_candidate_lines on the contours of the reference's stills against the oracle's refine_lines and against the
corners the reference's goldens hold (testdata/board/expected.yml, testdata/single/expected.yml)."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests.util import GOLDEN, load_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_candidates(gray):
    from oracle import orc

    o = orc.Oracle()
    o.detect_raw(gray)
    return [c for c in o.candidates(with_contour=True) if c["id"] >= 0]


@pytest.mark.parametrize("case,with_cam", [("board", False), ("single", True), ("chessboard", False)])
def test_refine_candidate_lines_equals_oracle_and_goldens(case, with_cam):
    from aruco_amd import capi
    from oracle import orc

    gray, doc = load_case(case)
    K, dist = (doc["intrinsics"]["K"], doc["intrinsics"]["dist"]) if with_cam else (None, None)
    cands = oracle_candidates(gray)
    assert len(cands) == len(doc["markers"])
    gold = {m["id"]: np.array(m["corners"], np.float64) for m in doc["markers"]}
    h = capi.Handle(640, 480, max_batch=1)
    try:
        for c in cands:
            got = h.refine_candidate_lines(c["contour"], c["quad0"], K=K, dist=dist)
            ref = orc.refine_lines(c["contour"], c["quad0"], K=K, dist=dist)
            assert np.max(np.abs(got - ref)) <= 1e-4 * np.max(np.abs(ref))                 # north_star: corners 1e-4 relative
            rot = np.roll(got, c["nrot"], axis=0)                                         # std::rotate(begin, begin + 4 - nRot, end), :364-366
            assert np.max(np.abs(rot - gold[c["id"]])) < 1e-3                              # pixels, against the reference's own file
        # the detector still works after the stage call borrowed its lists
        ids = [int(m["id"]) for m in h.detect(gray)]
        assert ids == [m["id"] for m in doc["markers"]]
    finally:
        h.close()


def test_refine_candidate_lines_rejects_bad_input():
    from aruco_amd import capi

    h = capi.Handle(640, 480, max_batch=1)
    try:
        sq = np.array([[10, 10], [20, 10], [20, 20], [10, 20]], np.int32)
        with pytest.raises(capi.ArucoHipError) as e:
            h.refine_candidate_lines(np.array([[-1, 5], [3, 4]], np.int32), sq.astype(np.float32))
        assert e.value.code == capi.E_INVALID
        with pytest.raises(capi.ArucoHipError) as e:
            h.refine_candidate_lines(np.zeros((0, 2), np.int32), sq.astype(np.float32))
        assert e.value.code == capi.E_INVALID
    finally:
        h.close()


def test_shim_refine_candidate_lines(tmp_path):
    """The shim's MarkerDetector::MarkerCandidate + refineCandidateLines(MarkerCandidate&, K, dist) through a C++ caller: the contours of
    the board still come from the library's own stage inspection calls, the refined corners must be those detect() delivers."""
    from aruco_amd import build_library

    build_library()
    exe = tmp_path / "shim_refine"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_refine.cpp"), "-o", str(exe),
                    "-L" + os.path.join(ROOT, "aruco_amd"), "-larucohip", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.join(ROOT, "aruco_amd"),
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), os.path.join(GOLDEN, "board.pgm")], stdout=subprocess.PIPE, text=True, check=True)
    doc = json.load(open(os.path.join(GOLDEN, "board.json")))
    gold = {m["id"]: np.array(m["corners"]) for m in doc["markers"]}
    seen = 0
    for line in r.stdout.splitlines():
        if not line.startswith("refined "):
            continue
        v = line.split()
        mid, c = int(v[1]), np.array([float(x) for x in v[2:10]]).reshape(4, 2)
        assert np.max(np.abs(c - gold[mid])) < 1e-3
        seen += 1
    assert seen == len(gold)
    assert "isYPerpendicular 0 1" in r.stdout


@pytest.mark.parametrize("case", ["single", "board", "chessboard", "hrm", "synthetic_1080p"])
def test_single_frame_graph_path_returns_the_bytes_of_the_eager_path(case, monkeypatch):
    """arucohip_detect on one host frame (the reference's call shape, test/perf_tests.cpp:31-56): from the second call of a configuration on
    the ~20 dispatches of a frame are replayed from a captured hipGraph. Eager handle (ARUCOHIP_GRAPH=0 at creation) against the default
    handle, three calls each (eager, capture + launch, replay), on the reference's four stills and one 1080p bench frame (walkers with the
    fork to the side stream): identical bytes every time - and again after the parameters change (a new graph) and change back."""
    from aruco_amd import capi, synth

    K = dist = None
    msize = -1.0
    p = capi.default_params()
    dic = None
    if case == "synthetic_1080p":
        fr, _ = synth.make_stream(2, seed=4711, device="cpu")
        gray, other = fr[0].numpy(), fr[1].numpy()
    else:
        gray, doc = load_case(case)
        other = np.ascontiguousarray(gray[:, ::-1])
        K, dist, msize = doc["intrinsics"]["K"], doc["intrinsics"]["dist"], 1.0
        if case == "hrm":
            st, dic = doc["settings"], doc["dictionary"]
            p.thres_param1, p.thres_param2, p.min_size, p.max_size, p.warp_size = st["thres_param1"], st["thres_param2"], st["min_size"], st["max_size"], st["warp_size"]
            msize = st["marker_size"]
    hgt, wid = gray.shape
    monkeypatch.setenv("ARUCOHIP_GRAPH", "0")
    eager = capi.Handle(wid, hgt, max_batch=1, params=p)
    monkeypatch.delenv("ARUCOHIP_GRAPH")
    graphed = capi.Handle(wid, hgt, max_batch=1, params=p)
    try:
        if dic:
            eager.set_dictionary(dic["markers"], dic["tau0"])
            graphed.set_dictionary(dic["markers"], dic["tau0"])
        ref = eager.detect(gray, K=K, dist=dist, marker_size=msize)
        ref_other = eager.detect(other, K=K, dist=dist, marker_size=msize)
        assert len(ref) >= 6
        for _ in range(3):
            assert graphed.detect(gray, K=K, dist=dist, marker_size=msize).tobytes() == ref.tobytes()
        assert graphed.detect(other, K=K, dist=dist, marker_size=msize).tobytes() == ref_other.tobytes()     # another frame through the same graph
        assert graphed.thresholded(0, (hgt, wid)).tobytes() == eager.thresholded(0, (hgt, wid)).tobytes()  # getters address the graphed call
        q = graphed.get_params()
        # another configuration = another graph (the HRM settings' threshold block of 21 is above the SUBPIX window limit: CORNER_NONE there)
        q.corner_method = capi.CORNER_NONE if case == "hrm" else capi.CORNER_SUBPIX
        graphed.set_params(q), eager.set_params(q)
        ref2 = eager.detect(gray, K=K, dist=dist, marker_size=msize)
        for _ in range(3):
            assert graphed.detect(gray, K=K, dist=dist, marker_size=msize).tobytes() == ref2.tobytes()
        q.corner_method = capi.CORNER_LINES
        graphed.set_params(q)
        for _ in range(3):
            assert graphed.detect(gray, K=K, dist=dist, marker_size=msize).tobytes() == ref.tobytes()
        # without a camera: another configuration again
        assert graphed.detect(gray).tobytes() == graphed.detect(gray).tobytes() == graphed.detect(gray).tobytes()
    finally:
        eager.close(), graphed.close()
