#!/usr/bin/env python3
"""Tuning sweep: one set of 1024 synthetic frames, the library's environment knobs varied between timed loops."""
import itertools, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aruco_amd import capi, synth

dev = torch.device("cuda", 0)
N = 1024
fr, truth = synth.make_stream(N, seed=4711, device=dev)
out = torch.zeros((N, 64 * 96), dtype=torch.uint8, device=dev)
cnt = torch.zeros(N, dtype=torch.int32, device=dev)
h = capi.Handle(1920, 1080, max_batch=N)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
h.set_stream(s.cuda_stream)
res = []
import ast
SETTINGS = ast.literal_eval(os.environ.get("SWEEP", "[{}]"))   # list of {ENV_NAME: value} dicts
for setting in SETTINGS:
    for k in [k for k in os.environ if k.startswith("ARUCOHIP_")]:
        del os.environ[k]
    for k, v in setting.items():
        os.environ[k] = str(v)
    for _ in range(2):
        h.detect_batch_device(fr.data_ptr(), N, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
    h.batch_status()
    h.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        h.detect_batch_device(fr.data_ptr(), N, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    h.batch_status()
    kt = {k: round(v, 3) for k, v in h.kernel_times().items() if v > 0.01}
    h.enable_timing(False)
    res.append({"setting": setting, "fps": round(N / dt, 1), "markers": round(float(cnt.float().mean()), 3), "kernel_ms": kt})
    print(json.dumps(res[-1]), flush=True)
