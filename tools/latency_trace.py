#!/usr/bin/env python3
"""A few single-frame detect() calls for a per-dispatch rocprofv3 trace of one call (tools/trace_latency.sh):
   latency_trace.py [single|board|1080p]   (the reference's 640x480 stills or one 1080p frame of the bench stream)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aruco_amd import capi
from aruco_amd.fixtures import load_case
case = sys.argv[1] if len(sys.argv) > 1 else "single"
K = dist = None
msize = -1.0
if case == "1080p":
    from aruco_amd import synth
    fr, _ = synth.make_stream(1, seed=4711, device="cpu")
    g = fr[0].numpy()
else:
    g, doc = load_case(case)
    if len(sys.argv) > 2 and sys.argv[2] == "pose":   # as test/perf_tests.cpp calls it: camera parameters given, poses are part of the call
        K, dist, msize = doc["intrinsics"]["K"], doc["intrinsics"]["dist"], 1.0
if "pageable" not in sys.argv:
    pinned = torch.empty(g.shape, dtype=torch.uint8, pin_memory=True)
    pinned.copy_(torch.from_numpy(g))
    g = pinned.numpy()
h = capi.Handle(g.shape[1], g.shape[0], max_batch=1)
for _ in range(30):
    h.detect(g, K=K, dist=dist, marker_size=msize)
t0 = time.perf_counter()
for _ in range(200):
    m = h.detect(g, K=K, dist=dist, marker_size=msize)
print("case", case, "markers", len(m), "ms per call", (time.perf_counter() - t0) / 200 * 1e3)
