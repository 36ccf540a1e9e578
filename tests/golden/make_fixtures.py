#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's test DATA.

Run in the build container only (needs /root/reference/testdata). Outputs are data:
  * <name>.pgm          gray raster of the reference's test PNG, converted with the OpenCV-3
                        integer BGR2GRAY formula; the reference's tests
                        feed the colour PNG to MarkerDetector::detect, which converts first
                        (src/markerdetector.cpp:307-310).
  * <name>.json         expected markers / board pose (testdata/*/expected.yml), intrinsics as the
                        float32 values CameraParameters::readFromXMLFile keeps
                        (src/cameraparameters.cpp:203-219: K -> f32, first 5 dist coeffs -> f32)
                        and the board configuration (board_pix.yml / chessboardinfo_pix.yml).
  * board_gl.json       testdata/board/expected_gl.yml (GL matrices; pins per-marker poses).
  * hrm.pgm / hrm.json  testdata/hrm (highly reliable markers, test/core_tests.cpp:310-353): frame, expected markers,
                        intrinsics (resized to the frame like the test does), the 4x4 dictionary d4x4_100.yml
                        (marker bit strings, tau0) and the detector settings the test applies.
  * hrm_dictionaries.json  the reference's larger dictionaries testdata/hrm/dictionaries/d5x5_100 .. d8x8_100.yml (marker bit
                        strings + tau0), which its HRM apps load; d4x4_100 is part of hrm.json.
  * create_marker.json  the CreateMarker goldens (test/core_tests.cpp:32-75, marker id 471, 500 px): the 7x7 cell matrix
                        decoded from testdata/board/marker-expected.png (every cell checked to be uniform 0 / 255) and, from
                        locked-marker-expected.png (750 px), the geometry of the locked-corner variant. Pins the bit layout
                        aruco_amd/synth.py::marker_bits draws.
No reference source text is copied; only test inputs and expected outputs.
"""
import json
import os
import re
import sys

import numpy as np
import yaml
from PIL import Image

REF = "/root/reference/testdata"
OUT = os.path.dirname(os.path.abspath(__file__))


def bgr2gray_cv3(rgb: np.ndarray) -> np.ndarray:
    r = rgb[..., 0].astype(np.int64)
    g = rgb[..., 1].astype(np.int64)
    b = rgb[..., 2].astype(np.int64)
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def load_cv_yaml(path):
    txt = open(path).read()
    txt = txt.replace("%YAML:1.0", "")
    txt = txt.replace("!!opencv-matrix", "")
    # OpenCV flow maps write "id:985" without a space; YAML needs "id: 985"
    txt = re.sub(r"(\w):(\S)", r"\1: \2", txt)
    txt = txt.replace(".Nan", ".nan")
    return yaml.safe_load(txt)


def write_pgm(path, gray):
    h, w = gray.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h))
        f.write(gray.tobytes())


def intrinsics(path):
    y = load_cv_yaml(path)
    K = np.array(y["camera_matrix"]["data"], dtype=np.float64).astype(np.float32)
    d = np.array(y["distortion_coefficients"]["data"], dtype=np.float64).astype(np.float32)[:5]
    return {
        "width": int(y["image_width"]),
        "height": int(y["image_height"]),
        # float32 values, stored as exact doubles
        "K": [float(v) for v in K],
        "dist": [float(v) for v in d],
    }


def markers(lst):
    out = []
    for m in lst:
        e = {"id": int(m["id"]), "corners": [[float(a), float(b)] for a, b in m["corners"]]}
        if "Rvec" in m:
            e["Rvec"] = [float(v) for v in m["Rvec"]]
            e["Tvec"] = [float(v) for v in m["Tvec"]]
        out.append(e)
    return out


def board_conf(path):
    y = load_cv_yaml(path)
    return {
        "info_type": int(y["aruco_bc_mInfoType"]),
        "ids": [int(m["id"]) for m in y["aruco_bc_markers"]],
        "obj": [[[float(v) for v in c] for c in m["corners"]] for m in y["aruco_bc_markers"]],
    }


def main():
    if not os.path.isdir(REF):
        sys.exit("reference testdata not available; fixtures are already committed")
    jobs = {
        "single": ("single/image-test.png", "single/expected.yml", "single/intrinsics.yml", None),
        "board": ("board/image-test.png", "board/expected.yml", "board/intrinsics.yml", "board/board_pix.yml"),
        "chessboard": ("chessboard/chessboard_frame.png", "chessboard/expected.yml",
                       "chessboard/intrinsics.yml", "chessboard/chessboardinfo_pix.yml"),
    }
    for name, (png, exp, intr, bc) in jobs.items():
        rgb = np.asarray(Image.open(os.path.join(REF, png)).convert("RGB"))
        write_pgm(os.path.join(OUT, name + ".pgm"), bgr2gray_cv3(rgb))
        y = load_cv_yaml(os.path.join(REF, exp))
        doc = {"source_png": png, "intrinsics": intrinsics(os.path.join(REF, intr))}
        if "Markers" in y:
            doc["markers"] = markers(y["Markers"])
        else:
            b = y["Board"]
            doc["board"] = {"Rvec": [float(v) for v in b["Rvec"]], "Tvec": [float(v) for v in b["Tvec"]]}
            doc["markers"] = markers(b["Markers"])
        if bc:
            doc["board_conf"] = board_conf(os.path.join(REF, bc))
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(doc, f, indent=1)
    # highly reliable markers
    rgb = np.asarray(Image.open(os.path.join(REF, "hrm/image-test.png")).convert("RGB"))
    write_pgm(os.path.join(OUT, "hrm.pgm"), bgr2gray_cv3(rgb))
    y = load_cv_yaml(os.path.join(REF, "hrm/expected.yml"))
    d = load_cv_yaml(os.path.join(REF, "hrm/dictionaries/d4x4_100.yml"))
    doc = {"source_png": "hrm/image-test.png", "intrinsics": intrinsics(os.path.join(REF, "hrm/intrinsics.yml")),
           "markers": markers(y["Markers"]),
           "dictionary": {"n": int(d["markersize"]), "tau0": int(d["tau0"]),
                          "markers": [str(d["marker_%d" % i]).zfill(int(d["markersize"]) ** 2) for i in range(int(d["nmarkers"]))]},
           # test/core_tests.cpp:324-329
           "settings": {"thres_param1": 21, "thres_param2": 7, "corner_method": "LINES", "min_size": 0.005, "max_size": 0.5,
                        "warp_size": (int(d["markersize"]) + 2) * 8, "marker_size": 1.0}}
    with open(os.path.join(OUT, "hrm.json"), "w") as f:
        json.dump(doc, f, indent=1)
    # test/core_tests.cpp:355-382 (Aruco.RefineFail): a frame on which the LINES refinement once failed; the test only
    # requires that detection goes through (same dictionary and settings, warp size 48)
    rgb = np.asarray(Image.open(os.path.join(REF, "hrm/refine-fail.png")).convert("RGB"))
    write_pgm(os.path.join(OUT, "hrm_refine_fail.pgm"), bgr2gray_cv3(rgb))
    # the larger dictionaries the reference ships (row f1)
    dicts = {}
    for n in (5, 6, 7, 8):
        d = load_cv_yaml(os.path.join(REF, "hrm/dictionaries/d%dx%d_100.yml" % (n, n)))
        assert int(d["markersize"]) == n
        dicts["d%dx%d_100" % (n, n)] = {"n": n, "tau0": int(d["tau0"]),
                                        "markers": [str(d["marker_%d" % i]).zfill(n * n) for i in range(int(d["nmarkers"]))]}
    with open(os.path.join(OUT, "hrm_dictionaries.json"), "w") as f:
        json.dump(dicts, f, indent=1)
    # CreateMarker goldens (test/core_tests.cpp:32-75): id 471 at 500 px; 500 / 7 is not an integer, createMarkerImage draws
    # cell (y, x) at [y*swidth, (y+1)*swidth) with swidth = 500 / 7 = 71 (integer division) and leaves the rest black
    img = np.asarray(Image.open(os.path.join(REF, "board/marker-expected.png")).convert("L"))
    assert img.shape == (500, 500) and set(np.unique(img)) <= {0, 255}
    sw = 500 // 7
    cells = np.zeros((7, 7), int)
    for y in range(7):
        for x in range(7):
            blk = img[y * sw:(y + 1) * sw, x * sw:(x + 1) * sw]
            assert blk.min() == blk.max(), "cell (%d, %d) of marker-expected.png is not uniform" % (y, x)
            cells[y, x] = 1 if blk[0, 0] == 255 else 0
    assert img[7 * sw:, :].max() == 0 and img[:, 7 * sw:].max() == 0
    locked = np.asarray(Image.open(os.path.join(REF, "board/locked-marker-expected.png")).convert("L"))
    with open(os.path.join(OUT, "create_marker.json"), "w") as f:
        json.dump({"source": "testdata/board/marker-expected.png", "marker_id": 471, "pix_size": 500, "cell_px": sw,
                   "cells": cells.tolist(), "locked_size": list(locked.shape)}, f, indent=1)
    gl = load_cv_yaml(os.path.join(REF, "board/expected_gl.yml"))
    with open(os.path.join(OUT, "board_gl.json"), "w") as f:
        json.dump({"gldata": [[float(v) for v in row] for row in gl["gldata"]]}, f, indent=1)
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
