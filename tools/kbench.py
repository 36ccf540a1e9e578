#!/usr/bin/env python3
"""Kernel micro-benchmark: per-kernel hipEvent times and list fill levels for a batch of synthetic 1080p frames."""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aruco_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--tag", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
fr, truth = synth.make_stream(a.frames, seed=4711, device=dev)
h = capi.Handle(1920, 1080, max_batch=a.frames)
out = torch.zeros((a.frames, 64 * 96), dtype=torch.uint8, device=dev)
cnt = torch.zeros(a.frames, dtype=torch.int32, device=dev)
for i in range(2):
    h.detect_batch_device(fr.data_ptr(), a.frames, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
h.batch_status()
h.enable_timing(True)
for i in range(a.steps):
    h.detect_batch_device(fr.data_ptr(), a.frames, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
h.batch_status()
kt = h.kernel_times()
c = h.debug_counters()
tot = sum(kt.values())
print(json.dumps({"tag": a.tag, "frames": a.frames, "fps_device": round(a.frames / tot * 1e3, 1), "total_ms": round(tot, 3),
                  "kernel_ms": {k: round(v, 3) for k, v in kt.items()}, "per_frame": {k: v / a.frames for k, v in c.items()},
                  "markers": float(cnt.float().mean())}))
