"""the CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer. `make -C oracle SAN=1 san_driver` builds the oracle's
sources with -fsanitize=address,undefined -fno-sanitize-recover=all (any finding aborts the process) plus a small C++ driver; the reference's
stills - single, board, chessboard (with BoardDetector), hrm, the RefineFail frame, and board with the reprojection filter - go through it and
must (a) finish clean and (b) print what the -O2 library computes. The same flags for the host-only C++ / C callers of the shim and the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from tests.util import GOLDEN, load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


@pytest.fixture(scope="module")
def driver():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "SAN=1", "san_driver"], check=True)
    return os.path.join(ROOT, "oracle", "san_driver")


def write_case(path, doc, marker_size=1.0, params=None, hrm=False, board=False, repj=None, with_cam=True):
    with open(path, "w") as f:
        if with_cam:
            intr = doc["intrinsics"]
            f.write("K " + " ".join(repr(float(v)) for v in intr["K"]) + "\n")
            f.write("dist %d " % len(intr["dist"]) + " ".join(repr(float(v)) for v in intr["dist"]) + "\n")
        f.write("size %r\n" % float(marker_size))
        if params:
            f.write("params %r %r %r %r %d %d\n" % tuple(params))
        if hrm:
            d = doc["dictionary"]
            f.write("hrm %d %d %d\n" % (d["n"], d["tau0"], len(d["markers"])))
            for m in d["markers"]:
                f.write(m + "\n")
        if board:
            bc = doc["board_conf"]
            f.write("board %d %d\n" % (bc["info_type"], len(bc["ids"])))
            for i, o in zip(bc["ids"], bc["obj"]):
                f.write("%d %s\n" % (i, " ".join(repr(float(v)) for p in o for v in p)))
        if repj is not None:
            f.write("repj %r\n" % float(repj))


def run(driver, image, case):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver, image, case], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    markers, board = [], None
    for line in r.stdout.splitlines():
        v = line.split()
        if v[0] == "marker":
            markers.append({"id": int(v[1]), "corners": np.array(v[2:10], float).reshape(4, 2), "rvec": np.array(v[10:13], float), "tvec": np.array(v[13:16], float)})
        elif v[0] == "board":
            board = {"prob": float(v[1]), "has_pose": int(v[2]), "rvec": np.array(v[3:6], float), "tvec": np.array(v[6:9], float)}
    return markers, board


@pytest.mark.parametrize("name", ["single", "board", "chessboard"])
def test_golden_stills_run_clean_and_equal_the_library(driver, tmp_path, name):
    from oracle import orc
    gray, doc = load_case(name)
    case = str(tmp_path / "case.txt")
    write_case(case, doc, board="board_conf" in doc, repj=1.5 if name == "board" else None)
    markers, board = run(driver, os.path.join(GOLDEN, name + ".pgm"), case)
    intr = doc["intrinsics"]
    ref = orc.Oracle().detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
    assert [m["id"] for m in markers] == [m["id"] for m in ref] == [e["id"] for e in doc["markers"]]
    for a, b in zip(markers, ref):
        assert np.allclose(a["corners"], b["corners"], rtol=0, atol=1e-5)
        assert np.allclose(a["rvec"], b["rvec"], rtol=1e-9, atol=1e-12) and np.allclose(a["tvec"], b["tvec"], rtol=1e-9, atol=1e-12)
    if "board_conf" in doc:
        assert board is not None and board["has_pose"] == 1 and abs(board["prob"] - len(markers) / len(doc["board_conf"]["ids"])) < 1e-6


def test_hrm_and_refine_fail_frames_run_clean(driver, tmp_path):
    _, doc = load_case("hrm")
    st = doc["settings"]
    case = str(tmp_path / "hrm.txt")
    write_case(case, doc, marker_size=st["marker_size"], params=(st["thres_param1"], st["thres_param2"], st["min_size"], st["max_size"], st["warp_size"], 3), hrm=True)
    markers, _ = run(driver, os.path.join(GOLDEN, "hrm.pgm"), case)
    assert [m["id"] for m in markers] == [e["id"] for e in doc["markers"]]
    # Aruco.RefineFail (test/core_tests.cpp:355-382): the frame on which the LINES corner walk once broke
    case2 = str(tmp_path / "fail.txt")
    write_case(case2, doc, marker_size=1.0, params=(21, 7, 0.005, 0.5, 48, 3), hrm=True)
    markers, _ = run(driver, os.path.join(GOLDEN, "hrm_refine_fail.pgm"), case2)
    assert len(markers) >= 10 and all(np.isfinite(m["corners"]).all() for m in markers)


def test_other_corner_methods_run_clean(driver, tmp_path):
    """SUBPIX and HARRIS (cornerSubPix / SubPixelCorner restatements) on the single still, no camera."""
    _, doc = load_case("single")
    for method in (1, 2, 0):
        case = str(tmp_path / ("m%d.txt" % method))
        write_case(case, doc, marker_size=-1.0, params=(7, 7, 0.04, 0.5, 56, method), with_cam=False)
        markers, _ = run(driver, os.path.join(GOLDEN, "single.pgm"), case)
        assert [m["id"] for m in markers] == [e["id"] for e in doc["markers"]]


def test_host_only_callers_run_clean_under_the_sanitizers(tmp_path):
    """tests/cpp/shim_yaml.cpp (the shim's YAML readers: pure host code) and the host-only calls of tests/cpp/c_caller.c compiled with the
    same sanitizer flags. The library they link is not instrumented, the callers and the header-only shim are. Leak checking is off for these
    two: the HIP runtime the library pulls in keeps process-lifetime allocations."""
    from aruco_amd import build_library
    lib = build_library()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    link = [lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"]
    exe = str(tmp_path / "c_caller_san")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra"] + SAN + ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "c_caller.c"), "-o", exe] + link, check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.startswith("version 100") and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
    src = os.path.join(ROOT, "tests", "cpp", "shim_yaml.cpp")
    exe = str(tmp_path / "shim_yaml_san")
    subprocess.run(["g++", "-std=c++17", "-Wall"] + SAN + ["-I" + os.path.join(ROOT, "include"), src, "-o", exe] + link, check=True)
    import tests.test_shim_yaml as ty
    ty.run_shim_yaml(exe, tmp_path, env=env)
