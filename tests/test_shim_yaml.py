"""SURVEY §8 row f3 (input side): the shim reads the reference's YAML files — CameraParameters::readFromXMLFile
(src/cameraparameters.cpp:187-222) and BoardConfiguration::readFromFile (src/serialization.cpp:94-120). The files are
written here in OpenCV FileStorage's layout (wrapped lines, flow-style marker list, unrelated keys, a matrix full of
.Nan) from the values of the committed goldens and must parse back to exactly those values. Host only."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def wrap(vals, per=3):
    rows = [", ".join(repr(float(v)) if float(v) != int(v) else "%d." % int(v) for v in vals[i:i + per]) for i in range(0, len(vals), per)]
    return ",\n       ".join(rows)


def test_shim_yaml_readers(tmp_path):
    from aruco_amd import build_library
    build_library()
    exe = tmp_path / "shim_yaml"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_yaml.cpp"), "-o", str(exe),
                    "-L" + os.path.join(ROOT, "aruco_amd"), "-larucohip", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.join(ROOT, "aruco_amd"),
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    run_shim_yaml(exe, tmp_path)


def run_shim_yaml(exe, tmp_path, env=None):
    """Writes the YAML files, runs the built caller on them and checks what it printed (also used by the sanitizer build of the same
    caller, tests/test_oracle_sanitized.py)."""
    doc = json.load(open(os.path.join(GOLDEN, "board.json")))
    intr, bc = doc["intrinsics"], doc["board_conf"]
    with open(tmp_path / "intrinsics.yml", "w") as f:
        f.write('%%YAML:1.0\ncalibration_time: "dom 27 feb 2011 17:16:27 CET"\nnframes: 5\nimage_width: %d\nimage_height: %d\n' % (intr["width"], intr["height"]))
        f.write("board_width: 8\nsquare_size: 2.8999999165534973e-02\ncamera_matrix: !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [ %s ]\n" % wrap(intr["K"], 4))
        f.write("distortion_coefficients: !!opencv-matrix\n   rows: 5\n   cols: 1\n   dt: d\n   data: [ %s ]\n" % wrap(intr["dist"], 2))
        f.write("avg_reprojection_error: 8.4719362981201396e-01\nextrinsic_parameters: !!opencv-matrix\n   rows: 1\n   cols: 4\n   dt: f\n   data: [ 6.47e-33, .Nan, 128.,\n       0. ]\n")
    with open(tmp_path / "board.yml", "w") as f:
        f.write("%%YAML:1.0\naruco_bc_nmarkers: %d\naruco_bc_mInfoType: %d\naruco_bc_markers:\n" % (len(bc["ids"]), bc["info_type"]))
        for i, o in zip(bc["ids"], bc["obj"]):
            c = ["[ %s ]" % ", ".join("%d." % int(v) if float(v) == int(v) else repr(float(v)) for v in p) for p in o]
            f.write("   - { id:%d, corners:[ %s, %s, [\n       %s ], %s ] }\n" % (i, c[0], c[1], c[2][2:-2], c[3]))
    bad = []
    for k, text in enumerate(["", "%YAML:1.0\nimage_width: 640\nimage_height: 480\ncamera_matrix: !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [ 1., 2. ]\n",
                              "%YAML:1.0\naruco_bc_nmarkers: 2\naruco_bc_mInfoType: 0\naruco_bc_markers:\n   - { id:1, corners:[ [ 0., 0., 0. ], [ 1., 0., 0. ] ] }\n",
                              "%YAML:1.0\nnmarkers: 2\nmarkersize: 4\ntau0: 4\nmarker_0: \"1011\"\nmarker_1: \"1001010011000100\"\n"]):
        path = tmp_path / ("bad%d.yml" % k)
        path.write_text(text)
        bad.append(str(path))
    bad.append(str(tmp_path / "does_not_exist.yml"))
    r = subprocess.run([str(exe), str(tmp_path / "intrinsics.yml"), str(tmp_path / "board.yml")] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    out = r.stdout.splitlines()
    assert [int(v) for v in out[0].split()] == [intr["width"], intr["height"]]
    assert np.array_equal(np.array(out[1].split(), np.float32), np.array(intr["K"], np.float32))
    assert np.array_equal(np.array(out[2].split(), np.float32), np.array(intr["dist"], np.float32))
    assert [int(v) for v in out[3].split()] == [bc["info_type"], len(bc["ids"])]
    for line, i, o in zip(out[4:4 + len(bc["ids"])], bc["ids"], bc["obj"]):
        v = line.split()
        assert int(v[0]) == i
        assert np.array_equal(np.array(v[1:], np.float32), np.array(o, np.float32).reshape(-1))
    assert out[4 + len(bc["ids"])] == "throws"
    # empty file, truncated matrix, marker with two corners, dictionary entry with too few bits, missing file:
    # all three readers reject all five (bit 1 = CameraParameters, 2 = BoardConfiguration, 4 = Dictionary)
    tail = out[5 + len(bc["ids"]):]
    assert tail == ["bad%d 7" % k for k in range(5)], tail
