#!/usr/bin/env python3
"""Experiment: N handles on N HIP streams, each detecting a slice of the batch concurrently (kernel overlap across streams)."""
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
res = {}
for ns in (1, 2, 4):
    per = a.frames // ns
    hs = [capi.Handle(1920, 1080, max_batch=per) for _ in range(ns)]
    ss = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    outs = [torch.zeros((per, 64 * 96), dtype=torch.uint8, device=dev) for _ in range(ns)]
    cnts = [torch.zeros(per, dtype=torch.int32, device=dev) for _ in range(ns)]
    for h, s in zip(hs, ss):
        h.set_stream(s.cuda_stream)
    def step():
        for i, h in enumerate(hs):
            h.detect_batch_device(fr[i * per:(i + 1) * per].data_ptr(), per, 1920, 1080, outs[i].data_ptr(), 64, cnts[i].data_ptr())
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    res[ns] = {"ms_per_step": round(dt * 1e3, 3), "fps": round(a.frames / dt, 1), "markers": float(sum(c.float().sum() for c in cnts)) / a.frames}
    for h in hs:
        h.batch_status()
    del hs
print(json.dumps(res))
