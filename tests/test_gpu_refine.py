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
