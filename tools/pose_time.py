#!/usr/bin/env python3
"""Latency of the batched solvePnP (arucohip_calculate_extrinsics) for a few and for many markers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aruco_amd import capi, synth
fr, truth = synth.make_stream(8, seed=4711, device="cuda")
h = capi.Handle(1920, 1080, max_batch=256)
K = [1400, 0, 960, 0, 1400, 540, 0, 0, 1]; d = [-0.10, 0.02, 1e-3, -5e-4, 0]
ms = np.concatenate(h.detect_batch_host(fr.cpu().numpy(), K=K, dist=d, marker_size=0.05))
print("markers", len(ms))
for n in (16, 256, 4096, 16384, 32768):
    m = np.resize(ms, n)
    h.calculate_extrinsics(m, K, d, 0.05)
    t0 = time.perf_counter()
    for _ in range(5):
        out = h.calculate_extrinsics(m, K, d, 0.05)
    dt = (time.perf_counter() - t0) / 5
    print(n, "markers: %.3f ms per call" % (dt * 1e3), int(out["has_pose"].sum()))
