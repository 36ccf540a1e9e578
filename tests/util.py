"""Shared helpers for the test-suite (fixture loading, comparison)."""
import json
import os

import numpy as np

from aruco_amd.fixtures import GOLDEN, load_case, read_pgm  # noqa: F401  (re-exported for the tests)


def rel_err(a, b):
    """max |a-b| / max(|b|) — vector-wise relative error (the north_star's '1e-4 relative')."""
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-12))


def load_hrm_dictionary(n):
    """(bit strings, tau0) of the reference's d<n>x<n>_100 dictionary (tests/golden/hrm_dictionaries.json; d4x4_100 lives in hrm.json)."""
    if n == 4:
        d = json.load(open(os.path.join(GOLDEN, "hrm.json")))["dictionary"]
    else:
        d = json.load(open(os.path.join(GOLDEN, "hrm_dictionaries.json")))["d%dx%d_100" % (n, n)]
    assert d["n"] == n
    return list(d["markers"]), int(d["tau0"])


def make_hrm_dictionary(n, count, tau, seed=3):
    """Random n x n dictionary (bit strings) whose markers differ from each other and from their own rotations in at least
    `tau` cells — a stand-in for the reference's d6x6 / d8x8 dictionaries."""
    def rots(code):
        m = np.array([c == "1" for c in code]).reshape(n, n)
        return ["".join("1" if v else "0" for v in np.rot90(m, -k).reshape(-1)) for k in range(4)]

    def dist(a, b):
        return sum(x != y for x, y in zip(a, b))

    rng = np.random.RandomState(seed)
    out = []
    while len(out) < count:
        c = "".join("1" if v else "0" for v in rng.rand(n * n) > 0.5)
        r = rots(c)
        if min(dist(r[0], r[k]) for k in (1, 2, 3)) < tau:
            continue
        if any(min(dist(o, x) for x in r) < tau for o in out):
            continue
        out.append(c)
    return out
