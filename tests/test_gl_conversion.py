"""OpenGL / Ogre conversions of the pose results against the reference's golden
(test/core_tests.cpp:230-283 Aruco.GL_Conversion <-> testdata/board/expected_gl.yml = tests/golden/board_gl.json).
Host arithmetic only, so it runs without a GPU; the reference compares with EXPECT_FLOAT_EQ (4 float ulps)."""
import json
import os

import numpy as np

from aruco_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def float_eq(a, b):
    """gtest's EXPECT_FLOAT_EQ on the values cast to float: within 4 ulps."""
    a32, b32 = np.float32(a), np.float32(b)
    return abs(float(a32) - float(b32)) <= 4 * np.spacing(max(abs(a32), abs(b32), np.float32(1e-30)))


def test_gl_conversion_golden():
    """gldata[0] = projection (0.5, 10), [1] = board modelview, [2..] = marker modelviews of testdata/board."""
    from oracle import orc
    from tests.util import load_case, rel_err
    gray, board = load_case("board")
    gl = json.load(open(os.path.join(GOLDEN, "board_gl.json")))["gldata"]
    intr = board["intrinsics"]
    K = np.array(intr["K"], np.float32).reshape(-1)
    size = (intr["width"], intr["height"])
    proj = capi.gl_projection(K, size, (gray.shape[1], gray.shape[0]), 0.5, 10)
    for j in range(16):   # exact inputs (intrinsics) -> the reference's own comparison
        assert float_eq(proj[j], gl[0][j]), (0, j, proj[j], gl[0][j])
    # The GL test detects WITH intrinsics (LINES refinement on undistorted points, per-marker solvePnP, marker size 1 from
    # test/test.h:17) and then runs the BoardDetector, so its poses are not those of board/expected.yml: they come from the
    # CPU restatement here and carry its 1e-4 relative pose tolerance.
    bc = board["board_conf"]
    ms = orc.Oracle().detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
    assert len(ms) == len(gl) - 2
    b = orc.board_detect(ms, bc["ids"], bc["obj"], bc["info_type"], intr["K"], intr["dist"], 1.0)
    assert rel_err(capi.gl_modelview(b["rvec"], b["tvec"]), gl[1]) < 1e-4
    for i, m in enumerate(ms):
        mv = capi.gl_modelview(m["rvec"], m["tvec"])
        assert rel_err(mv, gl[2 + i]) < 1e-4, (i, mv, gl[2 + i])


def test_ogre_conversions_consistent():
    """Ogre variants: projection = signed transpose of the GL one; the pose quaternion reproduces the axes the reference builds."""
    K = np.array([600, 0, 320, 0, 610, 240, 0, 0, 1], np.float32)
    p = capi.gl_projection(K, (640, 480), (1280, 960), 0.1, 100, invert=True).reshape(4, 4)
    o = capi.gl_projection(K, (640, 480), (1280, 960), 0.1, 100, invert=True, ogre=True).reshape(4, 4)
    sign = -np.ones((4, 4)); sign[:, 3] = 1
    assert np.allclose(o, sign * p.T, atol=0, rtol=0)
    rng = np.random.RandomState(5)
    for _ in range(50):
        rvec, tvec = rng.uniform(-3, 3, 3), rng.uniform(-2, 2, 3)
        pos, q = capi.ogre_pose(rvec, tvec)
        assert np.allclose(pos, [-tvec[0], -tvec[1], tvec[2]])
        assert abs(np.linalg.norm(q) - 1) < 1e-12
        th = np.linalg.norm(rvec); k = rvec / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx
        x = np.array([-R[0, 0], -R[1, 0], R[2, 0]]); y = np.array([-R[0, 1], -R[1, 1], R[2, 1]])
        A = np.stack([x, y, np.cross(x, y)], axis=1)
        w, qx, qy, qz = q
        Rq = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * w), 2 * (qx * qz + qy * w)],
                       [2 * (qx * qy + qz * w), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * w)],
                       [2 * (qx * qz - qy * w), 2 * (qy * qz + qx * w), 1 - 2 * (qx * qx + qy * qy)]])
        assert np.allclose(Rq, A, atol=1e-9)
        m = capi.gl_modelview(rvec, tvec).reshape(4, 4).T          # column-major -> rows
        assert np.allclose(m[:3, :3], R * np.array([[1], [1], [-1]]), atol=1e-12)
        assert np.allclose(m[:, 3], [tvec[0], tvec[1], -tvec[2], 1])


def test_shim_gl_methods(tmp_path):
    """Marker / Board / CameraParameters methods of the C++ shim reach the same entry points (host only, no GPU)."""
    import subprocess
    from aruco_amd import build_library
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build_library()
    exe = tmp_path / "shim_gl"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "shim_gl.cpp"), "-o", str(exe),
                    "-L" + os.path.join(root, "aruco_amd"), "-larucohip", "-L/opt/rocm/lib", "-Wl,-rpath," + os.path.join(root, "aruco_amd"),
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout.strip().splitlines()
    rvec, tvec = [0.1, -0.2, 0.3], [1, 2, 3]
    K = np.array([600, 0, 320, 0, 610, 240, 0, 0, 1], np.float32)
    assert np.array_equal(np.array(out[0].split(), float), capi.gl_modelview(rvec, tvec))
    pos, q = capi.ogre_pose(rvec, tvec)
    assert np.array_equal(np.array(out[1].split(), float), np.concatenate([pos, q]))
    assert np.array_equal(np.array(out[2].split(), float), capi.gl_projection(K, (640, 480), (640, 480), 0.5, 10))
    assert np.array_equal(np.array(out[3].split(), float), capi.gl_projection(K, (640, 480), (640, 480), 0.5, 10, invert=True, ogre=True))
    assert out[4] == "throws"
