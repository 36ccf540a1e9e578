"""Parity at BASELINE.json's full size: the 1024-frame 1920x1080 stream of config 2 / 3 (seed 4711), every frame against
the oracle (worker threads: the oracle's C calls release the GIL), plus the size-independent properties of the stream:
detected ids are rendered ids, corners sit on the rendering homography, a second pass and a pipelined pass (three batches
in flight) return the same bytes, and the frames of a permuted batch keep their results."""
import concurrent.futures as cf
import os

import numpy as np
import pytest

from tests.util import rel_err

pytestmark = pytest.mark.gpu

N = 1024
CAP = 64


@pytest.fixture(scope="module")
def env():
    import torch
    from aruco_amd import capi, synth
    from oracle import orc

    assert torch.cuda.is_available()
    capi.load()
    fr, truth = synth.make_stream(N, seed=4711, device="cuda")
    return {"capi": capi, "orc": orc, "torch": torch, "frames": fr, "truth": truth}


def _run(env, h, frames, K=None, dist=None, marker_size=-1.0):
    torch, capi = env["torch"], env["capi"]
    n = frames.shape[0]
    out = torch.zeros((n, CAP * 96), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    h.detect_batch_device(frames.data_ptr(), n, 1920, 1080, out.data_ptr(), CAP, cnt.data_ptr(), K=K, dist=dist, marker_size=marker_size)
    h.batch_status()
    torch.cuda.synchronize()
    arr = np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(n, CAP)
    return arr, cnt.cpu().numpy()


def test_all_1024_frames_equal_the_oracle_with_pose(env):
    """Config 3 on the whole stream: ids / order exact, corners and rvec / tvec <= 1e-4 relative on every frame."""
    capi, orc = env["capi"], env["orc"]
    K = [1400, 0, 960, 0, 1400, 540, 0, 0, 1]
    dist = [-0.10, 0.02, 1e-3, -5e-4, 0]
    h = capi.Handle(1920, 1080, max_batch=N)
    try:
        arr, cnt = _run(env, h, env["frames"], K=K, dist=dist, marker_size=0.05)
        arr2, cnt2 = _run(env, h, env["frames"], K=K, dist=dist, marker_size=0.05)
        assert np.array_equal(cnt, cnt2)
        for f in range(N):                                   # idempotence: same bytes from a second pass
            assert arr[f, :cnt[f]].tobytes() == arr2[f, :cnt2[f]].tobytes()
    finally:
        h.close()
    host = env["frames"].cpu().numpy()

    def ref(f):
        return orc.Oracle().detect(host[f], K=K, dist=dist, marker_size=0.05)

    with cf.ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
        refs = list(ex.map(ref, range(N)))
    total = 0
    for f in range(N):
        got, exp = arr[f, :cnt[f]], refs[f]
        assert [int(m["id"]) for m in got] == [m["id"] for m in exp], f
        for a, b in zip(got, exp):
            ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float).reshape(4, 2)
            assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < 1e-4, f      # north_star: 1e-4 relative
            assert int(a["has_pose"]) == 1
            assert rel_err(a["rvec"], b["rvec"]) < 1e-4 and rel_err(a["tvec"], b["tvec"]) < 1e-4, f
        total += len(got)
    assert total > 19 * N                                    # ~20 markers per frame are found
    # ground truth of the renderer: ids rendered, corners within 1.5 px of the rendering homography
    for f in range(N):
        tq = {t["id"]: t["quad"] for t in env["truth"][f]}
        for m in arr[f, :cnt[f]]:
            assert int(m["id"]) in tq, f
            c = np.asarray(m["corners"], float).reshape(4, 2)
            q = tq[int(m["id"])]
            assert min(np.max(np.linalg.norm(np.roll(q, k, axis=0) - c, axis=1)) for k in range(4)) < 1.5, f


def test_pipelined_and_permuted_batches_keep_every_frames_bytes(env):
    """Config 2 (no pose): three 1024-frame batches in flight return what one batch at a time returns, and a batch of the
    same frames in another order returns the same bytes per frame (frames are independent units)."""
    torch, capi = env["torch"], env["capi"]
    fr = env["frames"]
    h = capi.Handle(1920, 1080, max_batch=N)
    try:
        base, cnt = _run(env, h, fr)
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).cuda()
        shuffled = fr[perm].contiguous()
        ev = torch.cuda.Event()
        ev.record()                                          # behind the gather that writes `shuffled` on torch's stream
        h.wait_event(ev.cuda_event)                          # include/arucohip.h: frames from another stream are ordered by an event
        arr_p, cnt_p = _run(env, h, shuffled)
        pi = perm.cpu().numpy()
        for j in range(N):
            assert cnt_p[j] == cnt[pi[j]]
            assert arr_p[j, :cnt_p[j]].tobytes() == base[pi[j], :cnt[pi[j]]].tobytes()
        h.set_pipeline_depth(3)
        outs = [torch.zeros((N, CAP * 96), dtype=torch.uint8, device="cuda") for _ in range(3)]
        cnts = [torch.zeros(N, dtype=torch.int32, device="cuda") for _ in range(3)]
        srcs = [fr, shuffled, fr]
        tickets = [h.submit_device(srcs[i].data_ptr(), N, 1920, 1080, outs[i].data_ptr(), CAP, cnts[i].data_ptr()) for i in range(3)]
        for t in tickets:
            h.wait(t)
        torch.cuda.synchronize()
        for i, (ref_a, ref_c) in enumerate(((base, cnt), (arr_p, cnt_p), (base, cnt))):
            c = cnts[i].cpu().numpy()
            a = np.frombuffer(outs[i].cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(N, CAP)
            assert np.array_equal(c, ref_c)
            for f in range(N):
                assert a[f, :c[f]].tobytes() == ref_a[f, :c[f]].tobytes(), (i, f)
    finally:
        h.close()


def test_all_128_4k_board_frames_equal_the_oracle(env):
    """Config 4 at bench size: 128 frames 3840x2160 of the 6x4 board (bench.py's stream, seed 4711), MarkerDetector on every frame
    and the batched BoardDetector pose against the oracle; the pose also against the rendering pose (ground truth)."""
    from tests.util import load_case
    capi, orc, torch = env["capi"], env["orc"], env["torch"]
    from aruco_amd import synth
    W, H, NB = 3840, 2160, 128
    _, doc = load_case("board")
    bc = doc["board_conf"]
    K = np.array(doc["intrinsics"]["K"], np.float32).reshape(3, 3)          # CameraParameters::resize (cameraparameters.cpp:173-178)
    K[0, 0] *= np.float32(W / 640.0); K[0, 2] *= np.float32(W / 640.0)
    K[1, 1] *= np.float32(H / 480.0); K[1, 2] *= np.float32(H / 480.0)
    Kf = K.reshape(-1)
    dist = [0.0] * 5
    frames, poses = synth.make_board_stream(NB, bc["ids"], bc["obj"], Kf, width=W, height=H, seed=4711, device="cuda")
    torch.cuda.synchronize()
    h = capi.Handle(W, H, max_batch=NB)
    try:
        out = torch.zeros((NB, CAP * 96), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(NB, dtype=torch.int32, device="cuda")
        h.detect_batch_device(frames.data_ptr(), NB, W, H, out.data_ptr(), CAP, cnt.data_ptr())
        h.batch_status()
        boards = h.board_detect_batch(NB, bc["ids"], bc["obj"], bc["info_type"], Kf, dist, 0.039)
        torch.cuda.synchronize()
        arr = np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(NB, CAP)
        n = cnt.cpu().numpy()
    finally:
        h.close()
    host = frames.cpu().numpy()

    def ref(f):
        m = orc.Oracle().detect(host[f])
        return m, orc.board_detect(m, bc["ids"], bc["obj"], bc["info_type"], Kf, dist, 0.039)

    with cf.ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
        refs = list(ex.map(ref, range(NB)))
    for f in range(NB):
        got, (exp, ob) = arr[f, :n[f]], refs[f]
        assert [int(m["id"]) for m in got] == [m["id"] for m in exp], f
        assert len(got) >= 20
        for a, b in zip(got, exp):
            ca, cb = np.asarray(a["corners"], float).reshape(4, 2), np.asarray(b["corners"], float).reshape(4, 2)
            assert np.max(np.abs(ca - cb) / np.maximum(np.abs(cb), 1.0)) < 1e-4, f          # 1e-4 relative (north_star)
        bb = boards[f]
        assert bb["has_pose"] == 1 and ob["has_pose"] == 1 and bb["n_markers"] == len(ob["markers"])
        assert abs(bb["prob"] - ob["prob"]) < 1e-6
        assert rel_err(bb["rvec"], ob["rvec"]) < 1e-4 and rel_err(bb["tvec"], ob["tvec"]) < 1e-4, f
        # ground truth: the pose the frame was rendered with (corner detection noise: 1 % of the distance is ample)
        rv, tv = poses[f]
        assert np.linalg.norm(bb["tvec"] - tv) < 0.01 * np.linalg.norm(tv), f
