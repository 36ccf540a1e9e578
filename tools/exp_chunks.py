#!/usr/bin/env python3
"""Experiment: library-side chunking (ARUCOHIP_STREAMS) with and without kernel timing events."""
import argparse, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aruco_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
fr, truth = synth.make_stream(a.frames, seed=4711, device=dev)
out = torch.zeros((a.frames, 64 * 96), dtype=torch.uint8, device=dev)
cnt = torch.zeros(a.frames, dtype=torch.int32, device=dev)
res = {}
for ns in (1, 2, 4):
    os.environ["ARUCOHIP_STREAMS"] = str(ns)
    h = capi.Handle(1920, 1080, max_batch=a.frames)
    for own in (False,):
        if not own:
            s = torch.cuda.Stream(device=dev)
            torch.cuda.set_stream(s)
            h.set_stream(s.cuda_stream)
        for timing in (False,):
            h.enable_timing(timing)
            for _ in range(2):
                h.detect_batch_device(fr.data_ptr(), a.frames, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
            h.batch_status()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                h.detect_batch_device(fr.data_ptr(), a.frames, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
            h.batch_status()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            res["streams%d_%s_timing%d" % (ns, "own" if own else "torch", timing)] = round(a.frames / dt, 1)
    del h
print(json.dumps(res))
