"""Readers for the committed fixtures (tests/golden: gray rasters + the reference's expected values as JSON). Data only —
used by the tests, by __graft_entry__.smoke() and by bench.py's board configuration and latency leg."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = map(int, f.readline().split())
        f.readline()
        return np.frombuffer(f.read(), np.uint8).reshape(h, w).copy()


def load_case(name):
    """(gray raster, document) of tests/golden/<name>.pgm / .json"""
    doc = json.load(open(os.path.join(GOLDEN, name + ".json")))
    gray = read_pgm(os.path.join(GOLDEN, name + ".pgm"))
    return gray, doc
