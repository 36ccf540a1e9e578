#!/usr/bin/env python3
"""A few single-frame detect() calls on the 640x480 'single' still, for a per-dispatch rocprofv3 trace of one call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aruco_amd import capi
from aruco_amd.fixtures import load_case
g, _ = load_case("single")
h = capi.Handle(640, 480, max_batch=1)
for _ in range(30):
    h.detect(g)
t0 = time.perf_counter()
for _ in range(200):
    h.detect(g)
print("ms per call", (time.perf_counter() - t0) / 200 * 1e3)
