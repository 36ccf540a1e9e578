"""Pins the CPU oracle (oracle/) on the reference's own golden vectors
(reference tests: test/core_tests.cpp:77-116 Single, :164-195 Board, :197-228 Multi).

ids / counts / corner order: exact.  Sub-pixel corners: <= 1e-3 px absolute (measured 2.5e-4; the reference
fits lines with a float32 SVD, the oracle in double).  rvec/tvec: <= 1e-4 relative (north_star tolerance).
"""
import numpy as np
import pytest

from oracle import orc
from tests.util import load_case, rel_err

CORNER_ABS_TOL = 1e-3
POSE_REL_TOL = 1e-4
HS = 0.5
OBJ = [[-HS, -HS, 0], [-HS, HS, 0], [HS, HS, 0], [HS, -HS, 0]]


def test_single_golden():
    gray, doc = load_case("single")
    intr = doc["intrinsics"]
    ms = orc.Oracle().detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
    exp = doc["markers"]
    assert [m["id"] for m in ms] == [e["id"] for e in exp]
    for m, e in zip(ms, exp):
        assert np.max(np.abs(m["corners"] - np.array(e["corners"]))) < CORNER_ABS_TOL
        assert rel_err(m["rvec"], e["Rvec"]) < POSE_REL_TOL
        assert rel_err(m["tvec"], e["Tvec"]) < POSE_REL_TOL
        # what the reference test asserts: centre, 4 float ULP
        c = m["corners"].mean(axis=0)
        ce = np.array(e["corners"], np.float32).mean(axis=0)
        assert np.allclose(c, ce, rtol=1e-5)


def test_single_pnp_only():
    """solvePnP restatement alone: golden corners in, golden pose out (pins A.9 to ~1e-11)."""
    _, doc = load_case("single")
    intr = doc["intrinsics"]
    for e in doc["markers"]:
        ok, r, t = orc.solve_pnp(OBJ, e["corners"], intr["K"], intr["dist"])
        assert ok
        assert rel_err(r, e["Rvec"]) < 1e-9
        assert rel_err(t, e["Tvec"]) < 1e-9


@pytest.mark.parametrize("name", ["board", "chessboard"])
def test_board_golden(name):
    gray, doc = load_case(name)
    intr, bc = doc["intrinsics"], doc["board_conf"]
    ms = orc.Oracle().detect(gray)  # no intrinsics: pure LINES
    b = orc.board_detect(ms, bc["ids"], bc["obj"], bc["info_type"], intr["K"], intr["dist"], 1.0)
    exp = doc["markers"]
    assert [m["id"] for m in b["markers"]] == [e["id"] for e in exp]
    for m, e in zip(b["markers"], exp):
        assert np.max(np.abs(m["corners"] - np.array(e["corners"]))) < CORNER_ABS_TOL
    assert rel_err(b["rvec"], doc["board"]["Rvec"]) < POSE_REL_TOL
    assert rel_err(b["tvec"], doc["board"]["Tvec"]) < POSE_REL_TOL
    assert abs(b["prob"] - len(exp) / len(bc["ids"])) < 1e-6


def test_board_pnp_only():
    """Board solvePnP on the golden corners (96 / 28 points) reproduces the golden board pose."""
    for name in ("board", "chessboard"):
        _, doc = load_case(name)
        intr, bc = doc["intrinsics"], doc["board_conf"]
        ms = [{"id": e["id"], "corners": e["corners"]} for e in doc["markers"]]
        b = orc.board_detect(ms, bc["ids"], bc["obj"], bc["info_type"], intr["K"], intr["dist"], 1.0)
        assert rel_err(b["rvec"], doc["board"]["Rvec"]) < 1e-6
        assert rel_err(b["tvec"], doc["board"]["Tvec"]) < 1e-6


def test_hrm_golden():
    """Reference test Aruco.HRM_Single (test/core_tests.cpp:310-353): highly reliable markers, dictionary d4x4_100, the
    detector settings of the test, per-marker poses. ids exact; the test itself only compares ids, centres and poses."""
    gray, doc = load_case("hrm")
    intr, st, dic = doc["intrinsics"], doc["settings"], doc["dictionary"]
    o = orc.Oracle(thres_p1=st["thres_param1"], thres_p2=st["thres_param2"], min_size=st["min_size"], max_size=st["max_size"],
                   warp_size=st["warp_size"])
    o.set_hrm_dictionary(dic["markers"], dic["tau0"])
    ms = o.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=st["marker_size"])
    exp = doc["markers"]
    assert [m["id"] for m in ms] == [e["id"] for e in exp]
    for m, e in zip(ms, exp):
        assert np.max(np.abs(m["corners"] - np.array(e["corners"]))) < CORNER_ABS_TOL
        assert rel_err(m["rvec"], e["Rvec"]) < POSE_REL_TOL
        assert rel_err(m["tvec"], e["Tvec"]) < POSE_REL_TOL


def test_refine_fail_frame():
    """Reference test Aruco.RefineFail (test/core_tests.cpp:355-382): a frame on which the LINES refinement once broke;
    the test only demands that detect() goes through. Same here, plus finite results."""
    from tests.util import read_pgm, GOLDEN
    import os
    gray = read_pgm(os.path.join(GOLDEN, "hrm_refine_fail.pgm"))
    _, doc = load_case("hrm")
    intr, dic = doc["intrinsics"], doc["dictionary"]
    o = orc.Oracle(thres_p1=21, thres_p2=7, min_size=0.005, max_size=0.5, warp_size=48)
    o.set_hrm_dictionary(dic["markers"], dic["tau0"])
    ms = o.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
    assert len(ms) >= 10
    for m in ms:
        assert np.isfinite(m["corners"]).all() and np.isfinite(m["rvec"]).all() and np.isfinite(m["tvec"]).all()


def test_refine_lines_as_a_stage_reproduces_the_golden_corners():
    """MarkerDetector::refineCandidateLines called on its own (src/markerdetector.h:280, .cpp:931-997): contour + integer quad of every
    decoded candidate of the board and single stills -> the corners of testdata/board/expected.yml / single/expected.yml (the latter with
    the undistort / distort round trip of :956-959, :989-991)."""
    for name, with_cam in (("board", False), ("single", True)):
        gray, doc = load_case(name)
        K, dist = (doc["intrinsics"]["K"], doc["intrinsics"]["dist"]) if with_cam else (None, None)
        o = orc.Oracle()
        o.detect_raw(gray)
        gold = {m["id"]: np.array(m["corners"]) for m in doc["markers"]}
        seen = 0
        for c in o.candidates(with_contour=True):
            if c["id"] < 0:
                continue
            r = orc.refine_lines(c["contour"], c["quad0"], K=K, dist=dist)
            assert np.max(np.abs(np.roll(r, c["nrot"], axis=0) - gold[c["id"]])) < CORNER_ABS_TOL
            seen += 1
        assert seen == len(gold)
