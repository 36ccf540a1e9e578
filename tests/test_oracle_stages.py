"""Known-answer and property tests of the oracle's stages (independent numpy maths, no GPU)."""
import os

import numpy as np
import pytest

from aruco_amd import synth
from oracle import orc


def np_adaptive(gray, b, c):
    r = b // 2
    g = np.pad(gray.astype(np.int64), r, mode="edge")
    cs = np.cumsum(np.cumsum(np.pad(g, ((1, 0), (1, 0))), 0), 1)
    h, w = gray.shape
    s = cs[b:b + h, b:b + w] - cs[0:h, b:b + w] - cs[b:b + h, 0:w] + cs[0:h, 0:w]
    mean = (s + (b * b) // 2) // (b * b)
    return np.where(gray.astype(np.int64) - mean <= -int(np.floor(c)), 255, 0).astype(np.uint8)


def test_adaptive_threshold_matches_integral_image():
    rng = np.random.RandomState(0)
    for shape in ((37, 53), (64, 64), (5, 9)):
        g = rng.randint(0, 256, size=shape).astype(np.uint8)
        for b, c in ((7, 7.0), (3, 1.5), (11, -2.0)):
            assert np.array_equal(orc.adaptive_threshold(g, b, c), np_adaptive(g, b, c))


def test_contours_known_answer_rectangle_with_hole():
    img = np.zeros((12, 14), np.uint8)
    img[2:9, 3:11] = 255
    img[4:7, 5:9] = 0
    cs = orc.find_contours(img)
    assert len(cs) == 2
    hole, outer = cs[0], cs[1]            # RETR_LIST: reverse discovery order
    assert outer["hole"] == 0 and hole["hole"] == 1
    assert tuple(outer["pts"][0]) == (3, 2) and tuple(outer["pts"][1]) == (3, 3)   # starts top-left, goes down first
    assert len(outer["pts"]) == 2 * (7 + 8) - 4
    assert tuple(hole["pts"][0]) == (4, 4)                                         # pixel left of the hole's first pixel
    # every contour point is a foreground pixel, consecutive points are 8-neighbours, the chain closes
    for c in cs:
        p = c["pts"]
        assert np.all(img[p[:, 1], p[:, 0]] == 255)
        d = np.abs(np.diff(np.vstack([p, p[:1]]), axis=0))
        assert d.max() == 1


def test_contours_edge_cases():
    assert orc.find_contours(np.zeros((8, 8), np.uint8)) == []
    full = np.full((6, 7), 255, np.uint8)
    cs = orc.find_contours(full)          # the 1-px frame is cleared: one border around the 4x5 interior
    assert len(cs) == 1 and len(cs[0]["pts"]) == 2 * (4 + 5) - 4
    single = np.zeros((5, 5), np.uint8)
    single[2, 2] = 255
    cs = orc.find_contours(single)
    assert len(cs) == 1 and len(cs[0]["pts"]) == 1


def test_approx_poly_square_and_line():
    sq = [(x, 0) for x in range(0, 40)] + [(40, y) for y in range(0, 40)] + [(x, 40) for x in range(40, 0, -1)] + [(0, y) for y in range(40, 0, -1)]
    out = orc.approx_poly(np.array(sq), 0.05 * len(sq))
    assert sorted(map(tuple, out)) == [(0, 0), (0, 40), (40, 0), (40, 40)]
    line = [(x, 5) for x in range(30)] + [(x, 5) for x in range(28, 0, -1)]
    assert len(orc.approx_poly(np.array(line), 2.0)) == 2


@pytest.mark.parametrize("mid", [0, 1, 341, 682, 1023, 101])
def test_decode_all_rotations(mid):
    bits = synth.marker_bits(mid)
    img = np.kron(bits * 200 + 20, np.ones((8, 8))).astype(np.uint8)   # 56x56 canonical patch
    ids = []
    for k in range(4):
        got, nrot = orc.fiducial_detect(np.rot90(img, k).copy())
        ids.append((got, nrot))
    # rotationally symmetric codes may match earlier; every rotation must decode to the same id
    assert all(g == mid for g, _ in ids), ids
    assert ids[0][1] == 0
    bad = img.copy()
    bad[0:8, 8:16] = 255                  # white border cell -> not a marker
    assert orc.fiducial_detect(bad)[0] == -1


def test_pnp_round_trip():
    rng = np.random.RandomState(1)
    K = np.array([[900.0, 0, 640], [0, 900.0, 360], [0, 0, 1]])
    dist = [-0.1, 0.02, 1e-3, -5e-4, 0.0]
    hs = 0.05
    obj = np.array([[-hs, -hs, 0], [-hs, hs, 0], [hs, hs, 0], [hs, -hs, 0]])
    for _ in range(10):
        rvec = rng.uniform(-0.6, 0.6, 3) + np.array([np.pi * 0.8, 0, 0])
        tvec = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.1, 0.1), rng.uniform(0.5, 1.5)])
        R = synth._rodrigues(rvec)
        p = (R @ obj.T).T + tvec
        x, y = p[:, 0] / p[:, 2], p[:, 1] / p[:, 2]
        r2 = x * x + y * y
        cd = 1 + dist[0] * r2 + dist[1] * r2 ** 2 + dist[4] * r2 ** 3
        xd = x * cd + 2 * dist[2] * x * y + dist[3] * (r2 + 2 * x * x)
        yd = y * cd + dist[2] * (r2 + 2 * y * y) + 2 * dist[3] * x * y
        img = np.stack([xd * 900 + 640, yd * 900 + 360], 1)
        ok, r, t = orc.solve_pnp(obj, img, K.reshape(-1), dist)
        assert ok
        assert np.allclose(synth._rodrigues(r), R, atol=2e-4) and np.allclose(t, tvec, rtol=2e-4, atol=1e-5)


def test_synthetic_frame_detected_by_oracle():
    fr, truth = synth.make_stream(1, width=960, height=540, seed=3, n_markers=5, device="cpu")
    ms = orc.Oracle().detect(fr[0].numpy())
    ids = sorted(m["id"] for m in ms)
    assert ids == sorted(t["id"] for t in truth[0])


def test_bgr2gray_matches_the_fixture_pipeline():
    """Row f3: the oracle's cv::cvtColor(BGR2GRAY) restatement equals the 14-bit formula tests/golden/make_fixtures.py used
    to turn the reference's PNGs into the committed gray rasters (whose detections match the reference's goldens)."""
    rng = np.random.RandomState(11)
    bgr = rng.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    exp = ((bgr[..., 0].astype(np.int64) * 1868 + bgr[..., 1].astype(np.int64) * 9617 + bgr[..., 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(orc.bgr2gray(bgr), exp)
    g = rng.randint(0, 256, size=(8, 9)).astype(np.uint8)
    assert np.array_equal(orc.bgr2gray(np.repeat(g[..., None], 3, axis=2)), g)   # equal channels reproduce the value


def test_hrm_synthetic_6x6_and_8x8():
    """Larger dictionaries of row f1, the reference's own (testdata/hrm/dictionaries/d5x5_100 ... d8x8_100.yml as committed in
    tests/golden/hrm_dictionaries.json): the restatement finds exactly the rendered markers of a synthetic frame
    (MarkerCode::getImg layout), in any orientation."""
    from aruco_amd import synth
    from tests.util import load_hrm_dictionary
    for n in (5, 6, 7, 8):
        D, tau = load_hrm_dictionary(n)
        fr, lay = synth.make_hrm_frame(D, width=1280, height=720, seed=7 + n, n_markers=10)
        o = orc.Oracle(warp_size=(n + 2) * 8)
        o.set_hrm_dictionary(D, tau)
        ms = o.detect(fr.numpy())
        assert [m["id"] for m in ms] == sorted(m["id"] for m in lay)


def test_marker_bit_layout_equals_the_create_marker_golden():
    """Aruco.CreateMarker (test/core_tests.cpp:32-75 <-> testdata/board/marker-expected.png, id 471 at 500 px): the 7x7 cell matrix
    decoded from that PNG (tests/golden/create_marker.json) is what the synthetic generator draws for id 471, and a frame
    rendered from the golden cells decodes to 471 in the restatement."""
    import json
    from tests.util import GOLDEN
    doc = json.load(open(os.path.join(GOLDEN, "create_marker.json")))
    cells = np.array(doc["cells"], np.uint8)
    assert doc["marker_id"] == 471 and cells.shape == (7, 7)
    assert np.array_equal(synth.marker_bits(471), cells)
    # the golden image itself, rebuilt from its cells at the test's size (500 px, 71-px cells, the remainder black), on a white sheet
    sw = doc["cell_px"]
    img = np.zeros((500, 500), np.uint8)
    img[:7 * sw, :7 * sw] = np.kron(cells, np.ones((sw, sw), np.uint8)) * 255
    frame = np.full((1000, 1400), 255, np.uint8)   # contour of ~2000 points < 0.5 * 1400 * 4
    frame[100:600, 200:700] = img
    ms = orc.Oracle().detect(frame)
    assert [m["id"] for m in ms] == [471]
