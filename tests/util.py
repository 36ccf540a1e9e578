"""Shared helpers for the test-suite (fixture loading, comparison)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = map(int, f.readline().split())
        f.readline()
        return np.frombuffer(f.read(), np.uint8).reshape(h, w).copy()


def load_case(name):
    doc = json.load(open(os.path.join(GOLDEN, name + ".json")))
    gray = read_pgm(os.path.join(GOLDEN, name + ".pgm"))
    return gray, doc


def rel_err(a, b):
    """max |a-b| / max(|b|) — vector-wise relative error (the north_star's '1e-4 relative')."""
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-12))
