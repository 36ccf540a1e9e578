import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aruco_amd import capi, synth
capi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CAP = 64
fr, truth = synth.make_stream(N, seed=4711, device="cuda")
def run(h, frames):
    n = frames.shape[0]
    out = torch.zeros((n, CAP * 96), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    h.detect_batch_device(frames.data_ptr(), n, 1920, 1080, out.data_ptr(), CAP, cnt.data_ptr())
    h.batch_status()
    torch.cuda.synchronize()
    return np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(n, CAP), cnt.cpu().numpy()
perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).cuda()
sh = fr[perm].contiguous()
pi = perm.cpu().numpy()
h = capi.Handle(1920, 1080, max_batch=N)
a, ca = run(h, fr)
b, cb = run(h, sh)
bad = [j for j in range(N) if cb[j] != ca[pi[j]] or b[j, :cb[j]].tobytes() != a[pi[j], :ca[pi[j]]].tobytes()]
print("same handle, natural then permuted: mismatching slots", len(bad), bad[:20], [(int(cb[j]), int(ca[pi[j]])) for j in bad[:20]])
h.close()
h2 = capi.Handle(1920, 1080, max_batch=N)
b2, cb2 = run(h2, sh)
bad2 = [j for j in range(N) if cb2[j] != ca[pi[j]] or b2[j, :cb2[j]].tobytes() != a[pi[j], :ca[pi[j]]].tobytes()]
print("fresh handle, permuted: mismatching slots", len(bad2), bad2[:20], [(int(cb2[j]), int(ca[pi[j]])) for j in bad2[:20]])
b3, cb3 = run(h2, sh)
bad3 = [j for j in range(N) if cb3[j] != cb2[j] or b3[j, :cb3[j]].tobytes() != b2[j, :cb2[j]].tobytes()]
print("fresh handle, permuted twice: differing slots", len(bad3), bad3[:20])
a4, ca4 = run(h2, fr)
bad4 = [j for j in range(N) if ca4[j] != ca[j] or a4[j, :ca4[j]].tobytes() != a[j, :ca[j]].tobytes()]
print("then natural again: differing", len(bad4), bad4[:20], [(int(ca4[j]), int(ca[j])) for j in bad4[:20]])
for mode in ("ARUCOHIP_CAND_SPARSE", "ARUCOHIP_THRESHOLD_WIDE"):
    os.environ[mode] = "0"
    h3 = capi.Handle(1920, 1080, max_batch=N)
    run(h3, fr)
    b5, cb5 = run(h3, sh)
    bad5 = [j for j in range(N) if cb5[j] != ca[pi[j]] or b5[j, :cb5[j]].tobytes() != a[pi[j], :ca[pi[j]]].tobytes()]
    print(mode, "=0: natural then permuted mismatching", len(bad5), bad5[:10])
    h3.close()
    os.environ.pop(mode)
