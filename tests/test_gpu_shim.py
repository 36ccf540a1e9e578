"""The reference-shaped C++ API (include/aruco_hip_shim.hpp) driven by a C++ caller, config 1 of BASELINE.json:
aruco_simple on testdata/single (committed gray raster) must reproduce the reference's expected.yml."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from tests.util import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from aruco_amd import build_library

    build_library()
    out = tmp_path_factory.mktemp("shim") / "aruco_simple"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "aruco_simple.cpp"), "-o", str(out),
                    "-L" + os.path.join(ROOT, "aruco_amd"), "-larucohip", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.join(ROOT, "aruco_amd"),
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return str(out)


def write_intrinsics(path, intr):
    with open(path, "w") as f:
        f.write("%d %d\n" % (intr["width"], intr["height"]))
        f.write(" ".join(repr(v) for v in intr["K"]) + "\n")
        f.write(" ".join(repr(v) for v in intr["dist"]) + "\n")


def parse_markers(text):
    out = []
    for line in text.splitlines():
        m = re.match(r"^(\d+)=(.*)$", line)
        if not m:
            continue
        nums = [float(v) for v in re.findall(r"-?\d+\.?\d*(?:e[-+]?\d+)?", m.group(2))]
        d = {"id": int(m.group(1)), "corners": np.array(nums[:8]).reshape(4, 2)}
        if len(nums) >= 14:
            d["tvec"], d["rvec"] = np.array(nums[8:11]), np.array(nums[11:14])
        out.append(d)
    return out


def test_aruco_simple_single(exe, tmp_path):
    doc = json.load(open(os.path.join(GOLDEN, "single.json")))
    intr = tmp_path / "intr.txt"
    write_intrinsics(intr, doc["intrinsics"])
    r = subprocess.run([exe, os.path.join(GOLDEN, "single.pgm"), str(intr), "1.0"], stdout=subprocess.PIPE, text=True, check=True)
    got = parse_markers(r.stdout)
    exp = doc["markers"]
    assert [g["id"] for g in got] == [e["id"] for e in exp]
    for g, e in zip(got, exp):
        assert np.max(np.abs(g["corners"] - np.array(e["corners"]))) < 1e-3
        assert rel_err(g["rvec"], e["Rvec"]) < 1e-4 and rel_err(g["tvec"], e["Tvec"]) < 1e-4
    assert "setWarpSize(5) rejected" in r.stdout          # CV_Assert(val >= 10) -> cv::Exception
    assert "thres=640x480" in r.stdout


def test_aruco_simple_board(exe, tmp_path):
    doc = json.load(open(os.path.join(GOLDEN, "board.json")))
    intr = tmp_path / "intr.txt"
    write_intrinsics(intr, doc["intrinsics"])
    bc = doc["board_conf"]
    with open(tmp_path / "board.txt", "w") as f:
        f.write("%d %d\n" % (bc["info_type"], len(bc["ids"])))
        for i, o in zip(bc["ids"], bc["obj"]):
            f.write("%d %s\n" % (i, " ".join(repr(v) for p in o for v in p)))
    r = subprocess.run([exe, os.path.join(GOLDEN, "board.pgm"), str(intr), "1.0", str(tmp_path / "board.txt")], stdout=subprocess.PIPE, text=True,
                       check=True)
    got = parse_markers(r.stdout)
    assert [g["id"] for g in got] == [e["id"] for e in doc["markers"]]
    m = re.search(r"board prob=(\S+) Rvec=(\S+) (\S+) (\S+) Tvec=(\S+) (\S+) (\S+)", r.stdout)
    assert m and abs(float(m.group(1)) - 1.0) < 1e-6
    assert rel_err([float(m.group(i)) for i in (2, 3, 4)], doc["board"]["Rvec"]) < 1e-4
    assert rel_err([float(m.group(i)) for i in (5, 6, 7)], doc["board"]["Tvec"]) < 1e-4


def test_shim_hrm(tmp_path):
    """Row f1 through the shim: Dictionary::fromFile on a file in the reference's format, HighlyReliableMarkers::
    loadDictionary, MarkerDetector::setMakerDetectorFunction(HighlyReliableMarkers::detect) — the calls of the reference's
    HRM_Single test — reproduce testdata/hrm/expected.yml."""
    from aruco_amd import build_library
    build_library()
    doc = json.load(open(os.path.join(GOLDEN, "hrm.json")))
    dic = doc["dictionary"]
    with open(tmp_path / "dict.yml", "w") as f:
        f.write("%%YAML:1.0\nnmarkers: %d\nmarkersize: %d\ntau0: %d\n" % (len(dic["markers"]), dic["n"], dic["tau0"]))
        for i, m in enumerate(dic["markers"]):
            f.write('marker_%d: "%s"\n' % (i, m))
    intr = tmp_path / "intr.txt"
    write_intrinsics(intr, doc["intrinsics"])
    exe = tmp_path / "shim_hrm"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_hrm.cpp"), "-o", str(exe),
                    "-L" + os.path.join(ROOT, "aruco_amd"), "-larucohip", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.join(ROOT, "aruco_amd"),
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), os.path.join(GOLDEN, "hrm.pgm"), str(tmp_path / "dict.yml"), str(intr)], stdout=subprocess.PIPE, text=True, check=True)
    got = parse_markers(r.stdout)
    exp = doc["markers"]
    assert [g["id"] for g in got] == [e["id"] for e in exp]
    for g, e in zip(got, exp):
        assert np.max(np.abs(g["corners"] - np.array(e["corners"]))) < 1e-3
        assert rel_err(g["rvec"], e["Rvec"]) < 1e-4 and rel_err(g["tvec"], e["Tvec"]) < 1e-4
    assert "fiducial=" in r.stdout


def test_shim_member_extrinsics_and_user_decoder(tmp_path):
    """Round 2 boundary rows: Marker::calculateExtrinsics (reference src/marker.h:77,85) against the golden poses of
    testdata/single, and setMakerDetectorFunction with the caller's own host decoder (src/markerdetector.h:65-78) against
    the device decoder — through a C++ caller of the shim."""
    from aruco_amd import build_library
    build_library()
    doc = json.load(open(os.path.join(GOLDEN, "single.json")))
    intr = tmp_path / "intr.txt"
    write_intrinsics(intr, doc["intrinsics"])
    exe = tmp_path / "shim_callbacks"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_callbacks.cpp"), "-o", str(exe),
                    "-L" + os.path.join(ROOT, "aruco_amd"), "-larucohip", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.join(ROOT, "aruco_amd"),
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), os.path.join(GOLDEN, "single.pgm"), str(intr)], stdout=subprocess.PIPE, text=True, check=True)
    lines = r.stdout.splitlines()
    exp = doc["markers"]
    got = parse_markers("\n".join(l[5:] for l in lines if l.startswith("extr ")))
    assert [g["id"] for g in got] == [e["id"] for e in exp]
    for g, e in zip(got, exp):
        assert rel_err(g["rvec"], e["Rvec"]) < 1e-4 and rel_err(g["tvec"], e["Tvec"]) < 1e-4   # north_star tolerance
    assert "invalid marker rejected" in r.stdout
    dev = [l[4:] for l in lines if l.startswith("dev ")]
    usr = [l[4:] for l in lines if l.startswith("usr ")]
    assert len(dev) == len(exp) and dev == usr            # identical ids, corner order, corners and poses
    m = re.search(r"decoder calls=(\d+) candidates=(\d+)", r.stdout)
    assert m and int(m.group(1)) == len(exp) + int(m.group(2))   # one call per candidate of detectRectangles
    assert "after reset: calls 0 markers %d" % len(exp) in r.stdout


def test_plain_c_caller_detects_the_single_still(tmp_path):
    """tests/cpp/c_caller.c (C99, no C++ runtime of its own) through arucohip_create / arucohip_detect on the reference's single still:
    the ids of testdata/single/expected.yml."""
    from aruco_amd import build_library

    lib = build_library()
    exe_c = str(tmp_path / "c_caller")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "c_caller.c"),
                    "-o", exe_c, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe_c, os.path.join(GOLDEN, "single.pgm")], stdout=subprocess.PIPE, text=True, check=True).stdout
    doc = json.load(open(os.path.join(GOLDEN, "single.json")))
    ids = [int(v) for v in out.strip().splitlines()[-1].split(":")[1].split()]
    assert ids == [m["id"] for m in doc["markers"]]
