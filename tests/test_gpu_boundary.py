"""GPU tests of the C-ABI boundary rows added in round 2: the caller's own decoder (plugin boundary, reference
src/markerdetector.h:65-78,243-245) and the batched / multi-device entry points."""
import ctypes as C

import numpy as np
import pytest

from tests.util import load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch  # noqa: F401
    from aruco_amd import capi, synth

    assert torch.cuda.is_available()
    capi.load()
    return {"capi": capi, "torch": torch, "synth": synth}


WORDS = np.array([[1, 0, 0, 0, 0], [1, 0, 1, 1, 1], [0, 1, 0, 0, 1], [0, 1, 1, 1, 0]], np.int32)


def numpy_5x5_decoder(patch):
    """Host decoder written for this test (not the oracle): Otsu by exhaustive between-class variance, 7x7 cell votes,
    four rotations against the Hamming words. Returns (id or -1, nRotations)."""
    n = patch.shape[0]
    hist = np.bincount(patch.reshape(-1), minlength=256).astype(np.float64)
    p = hist / hist.sum()
    q1 = np.cumsum(p)
    m = np.cumsum(p * np.arange(256))
    mu = m[-1]
    q2 = 1.0 - q1
    ok = (np.minimum(q1, q2) >= 1.1920929e-7) & (np.maximum(q1, q2) <= 1 - 1.1920929e-7)
    with np.errstate(divide="ignore", invalid="ignore"):
        sigma = np.where(ok, q1 * q2 * (m / q1 - (mu - m) / q2) ** 2, 0.0)
    thr = int(np.argmax(sigma)) if sigma.max() > 0 else 0
    sw = n // 7
    cells = (patch[:7 * sw, :7 * sw] > thr).reshape(7, sw, 7, sw).sum(axis=(1, 3)) > (sw * sw) // 2
    if cells[0].any() or cells[6].any() or cells[:, 0].any() or cells[:, 6].any():
        return -1, 0
    code = cells[1:6, 1:6].astype(np.int32)

    def dist(c):
        return int(sum(min(int((row != w).sum()) for w in WORDS) for row in c))

    best, nrot, keep = dist(code), 0, code
    cur = code
    for r in range(1, 4):
        cur = np.rot90(cur, -1)            # out(i, j) = in(n - j - 1, i)
        d = dist(cur)
        if d < best:
            best, nrot, keep = d, r, cur
    if best != 0:
        return -1, nrot
    ident = 0
    for y in range(5):
        ident = (ident << 2) | (int(keep[y, 1]) << 1) | int(keep[y, 3])
    return ident, nrot


@pytest.mark.parametrize("case", ["single", "board", "chessboard"])
def test_user_decoder_callback_equals_device_decoder(env, case):
    """arucohip_set_decoder_callback: device warps, host decodes, device continues; the marker bytes equal the device
    decoder's."""
    capi = env["capi"]
    gray, doc = load_case(case)
    intr = doc["intrinsics"]
    h = capi.Handle(640, 480, max_batch=1)
    try:
        ref = h.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
        ncand_total = len(h.debug_candidates(0)[1])
        calls = []
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_int))

        def cb(user, patch, size, nrot):
            a = np.ctypeslib.as_array(patch, shape=(size, size))
            ident, r = numpy_5x5_decoder(a.copy())
            a[:] = 0                        # the patch is scratch: a decoder may destroy it
            nrot[0] = r
            calls.append(ident)
            return ident

        fn = FN(cb)
        L = capi.load()
        L.arucohip_set_decoder_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        p = h.get_params()
        p.decoder_kind = 2
        with pytest.raises(capi.ArucoHipError):      # USER without a callback is refused when a frame arrives
            h.set_params(p)
            h.detect(gray)
        assert L.arucohip_set_decoder_callback(h.h, C.cast(fn, C.c_void_p), None) == 0
        h.set_params(p)
        got = h.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
        assert got.tobytes() == ref.tobytes()
        assert len(calls) == ncand_total and sorted(i for i in calls if i >= 0) == sorted(int(m["id"]) for m in ref)
        assert [int(m["id"]) for m in got] == [e["id"] for e in doc["markers"]]
        # removing the callback restores the device decoder
        assert L.arucohip_set_decoder_callback(h.h, None, None) == 0
        assert h.get_params().decoder_kind == 0
        assert h.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0).tobytes() == ref.tobytes()
    finally:
        h.close()


def test_user_decoder_on_a_batch(env):
    capi, torch = env["capi"], env["torch"]
    frames, truth = env["synth"].make_stream(3, width=1920, height=1080, seed=23, device="cuda")
    fr = frames.cpu().numpy()
    h = capi.Handle(1920, 1080, max_batch=3)
    try:
        ref = h.detect_batch_host(fr)
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_int))

        def cb(user, patch, size, nrot):
            ident, r = numpy_5x5_decoder(np.ctypeslib.as_array(patch, shape=(size, size)))
            nrot[0] = r
            return ident

        fn = FN(cb)
        L = capi.load()
        L.arucohip_set_decoder_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        assert L.arucohip_set_decoder_callback(h.h, C.cast(fn, C.c_void_p), None) == 0
        p = h.get_params()
        p.decoder_kind = 2
        h.set_params(p)
        got = h.detect_batch_host(fr)
        assert sum(len(g) for g in got) >= 50
        for a, b in zip(got, ref):
            assert a.tobytes() == b.tobytes()
    finally:
        h.close()


@pytest.mark.parametrize("flags", [0, 1])
def test_mgpu_sharding_equals_single_handle(env, flags):
    """arucohip_mgpu_detect_batch / _detect_streams: frame f on slot f mod G, blocks gathered (host copy or device-to-device
    into the first device). On this one-GPU box the slots are three handles on device 0 — the sharding, the strided
    sub-batches, the ragged last round and the gather are the code that runs on G GPUs."""
    capi, torch = env["capi"], env["torch"]
    frames, truth = env["synth"].make_stream(7, width=1920, height=1080, seed=31, device="cuda")
    fr = frames.cpu().numpy()
    single = capi.Handle(1920, 1080, max_batch=7)
    try:
        ref = single.detect_batch_host(fr, cap=64)
    finally:
        single.close()
    mg = capi.MultiGpu([0, 0, 0], 1920, 1080, frames_per_device=3, cap=64, flags=flags)
    try:
        assert capi.load().arucohip_mgpu_size(mg.m) == 3
        got = mg.detect_batch_host(fr)                        # 7 frames over 3 slots: 3 + 2 + 2
        assert len(got) == 7
        for a, b in zip(got, ref):
            assert a.tobytes() == b.tobytes()
        assert sum(len(g) for g in got) >= 100
        with pytest.raises(capi.ArucoHipError):               # more frames than slots x frames per slot
            mg.detect_batch_host(np.zeros((10, 1080, 1920), np.uint8))
        # camera streams resident per slot (config 5): slot g gets frames [2g, 2g+1], the last slot a single frame
        ptrs = [frames[0].data_ptr(), frames[2].data_ptr(), frames[4].data_ptr()]
        res = mg.detect_streams(ptrs, [2, 2, 1], 1920, 1080)
        for g, cnt in enumerate([2, 2, 1]):
            for j in range(cnt):
                assert res[g][j].tobytes() == ref[2 * g + j].tobytes()
        # asynchronous form (round 3): two tickets outstanding, every slot keeps two batches in flight; results as the synchronous calls
        mg.set_depth(2)
        j0 = mg.submit_batch_host(fr)
        j1 = mg.submit_streams(ptrs, [2, 2, 1], 1920, 1080)
        with pytest.raises(capi.ArucoHipError) as e:            # a third ticket needs a wait first
            mg.submit_batch_host(fr)
        assert e.value.code == capi.E_CAPACITY
        got0 = mg.wait(j0)
        j2 = mg.submit_batch_host(fr[:5])                     # ragged: 2 + 2 + 1
        res1 = mg.wait(j1)
        got2 = mg.wait(j2)
        with pytest.raises(capi.ArucoHipError):               # a ticket is waited for once
            mg.wait(j2)
        for a, b in zip(got0, ref):
            assert a.tobytes() == b.tobytes()
        for a, b in zip(got2, ref[:5]):
            assert a.tobytes() == b.tobytes()
        for g, cnt in enumerate([2, 2, 1]):
            for j in range(cnt):
                assert res1[g][j].tobytes() == ref[2 * g + j].tobytes()
        mg.set_depth(1)
        for a, b in zip(mg.detect_batch_host(fr), ref):
            assert a.tobytes() == b.tobytes()
    finally:
        mg.close()


def test_mgpu_set_depth_is_transactional(env, monkeypatch):
    """arucohip_mgpu_set_depth when building the new lanes fails (injected behind the first slot, the way an out-of-memory lane would):
    the call reports the error and the detector keeps running at its previous depth; when the previous depth cannot be rebuilt either,
    every later call returns ARUCOHIP_E_HIP at once - no job is queued for threads that do not exist (round 3: arucohip_mgpu_wait hung)."""
    capi = env["capi"]
    frames, _ = env["synth"].make_stream(4, width=1920, height=1080, seed=31, device="cuda")
    fr = frames.cpu().numpy()
    mg = capi.MultiGpu([0, 0], 1920, 1080, frames_per_device=2, cap=64)
    try:
        ref = mg.detect_batch_host(fr)
        monkeypatch.setenv("ARUCOHIP_MGPU_INJECT", "fail_depth:3")
        with pytest.raises(capi.ArucoHipError) as e:
            mg.set_depth(3)
        assert e.value.code == capi.E_HIP and "previous depth 1 was restored" in str(e.value)
        for a, b in zip(mg.detect_batch_host(fr), ref):        # still works, at depth 1
            assert a.tobytes() == b.tobytes()
        monkeypatch.delenv("ARUCOHIP_MGPU_INJECT")
        mg.set_depth(3)                                        # and the same request succeeds once nothing fails
        j = [mg.submit_batch_host(fr) for _ in range(3)]
        for jj in j:
            for a, b in zip(mg.wait(jj), ref):
                assert a.tobytes() == b.tobytes()
        monkeypatch.setenv("ARUCOHIP_MGPU_INJECT", "fail_depth:2,3")
        with pytest.raises(capi.ArucoHipError) as e:
            mg.set_depth(2)                                    # neither 2 nor the previous 3 can be built
        assert "unusable" in str(e.value)
        with pytest.raises(capi.ArucoHipError) as e:
            mg.detect_batch_host(fr)                           # returns at once
        assert e.value.code == capi.E_HIP
        monkeypatch.delenv("ARUCOHIP_MGPU_INJECT")
        mg.set_depth(1)                                        # a later rebuild that succeeds repairs the detector
        for a, b in zip(mg.detect_batch_host(fr), ref):
            assert a.tobytes() == b.tobytes()
    finally:
        mg.close()


def test_mgpu_peer_gather_falls_back_to_the_host_gather(env, monkeypatch):
    """ARUCOHIP_MGPU_GATHER_PEER on devices that cannot reach the first one (hipDeviceCanAccessPeer false - injected here, the box has one
    GPU): the detector gathers through pinned host memory instead and says so; results are unchanged."""
    capi = env["capi"]
    frames, _ = env["synth"].make_stream(3, width=1920, height=1080, seed=33, device="cuda")
    fr = frames.cpu().numpy()
    mg = capi.MultiGpu([0, 0], 1920, 1080, frames_per_device=2, cap=64, flags=capi.MultiGpu.GATHER_PEER)
    try:
        assert mg.gather_mode() == capi.MultiGpu.GATHER_PEER   # slots on one device reach each other trivially
        ref = mg.detect_batch_host(fr)
    finally:
        mg.close()
    monkeypatch.setenv("ARUCOHIP_MGPU_INJECT", "nopeer")
    mg = capi.MultiGpu([0, 0], 1920, 1080, frames_per_device=2, cap=64, flags=capi.MultiGpu.GATHER_PEER)
    try:
        assert mg.gather_mode() == capi.MultiGpu.GATHER_HOST
        for a, b in zip(mg.detect_batch_host(fr), ref):
            assert a.tobytes() == b.tobytes()
    finally:
        mg.close()


def test_compact_markers_kernel_equals_the_host_packing(env):
    """arucohip_compact_markers (the block a rank contributes to the RCCL gather) against aruco_amd.dist.pack_block on the result
    arrays of a real batch, a block packed too small (overflow flag, counts intact) and counts that carry -1 / more than cap."""
    capi, torch = env["capi"], env["torch"]
    from aruco_amd import dist as adist
    frames, _ = env["synth"].make_stream(9, width=1920, height=1080, seed=17, device="cuda")
    CAPM = 64
    h = capi.Handle(1920, 1080, max_batch=9)
    try:
        out = torch.zeros((9, CAPM * 96), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(9, dtype=torch.int32, device="cuda")
        h.detect_batch_device(frames.data_ptr(), 9, 1920, 1080, out.data_ptr(), CAPM, cnt.data_ptr())
        h.batch_status()
    finally:
        h.close()
    total = int(cnt.sum().item())
    assert total > 150
    st = torch.cuda.Stream()
    for cap_total, counts in ((total + 7, cnt), (total - 30, cnt), (9 * CAPM, torch.tensor([3, -1, 200, 0, 5, 64, 1, 2, 70], dtype=torch.int32, device="cuda"))):
        dst = torch.full((capi.compact_bytes(9, cap_total),), 0xAB, dtype=torch.uint8, device="cuda")
        assert dst.numel() == adist.block_bytes(9, cap_total)
        capi.compact_markers(out.data_ptr(), counts.data_ptr(), 9, CAPM, dst.data_ptr(), cap_total, st.cuda_stream)
        st.synchronize()
        ref = adist.pack_block(out.cpu(), counts.cpu(), CAPM, cap_total)
        got = dst.cpu()
        used = min(int(counts.clamp(0, CAPM).sum().item()), cap_total)
        hb = adist.block_head_bytes(9)
        assert torch.equal(got[:16 + 36], ref[:16 + 36])                        # header + counts
        assert torch.equal(got[hb:hb + used * 96], ref[hb:hb + used * 96])      # the markers that fit
        c, fr_, ovf = adist.unpack_block(got, CAPM, capi.MARKER_DTYPE)
        assert ovf == (int(counts.clamp(0, CAPM).sum().item()) > cap_total)


def test_frames_from_another_stream_ordered_by_an_event(env):
    """The stream contract of device frames (include/arucohip.h): frames produced on another stream are ordered in front of the batch
    with arucohip_wait_event instead of a device-wide synchronise. The producer here is slow on purpose (a chain of full-frame kernels
    on its own stream in front of the permutation that writes the frames); without the event the batch would read frames that are still
    being written (round 2 saw 5 of 20 markers in exactly this situation)."""
    capi, torch = env["capi"], env["torch"]
    frames, truth = env["synth"].make_stream(48, width=1920, height=1080, seed=23, device="cuda")
    h = capi.Handle(1920, 1080, max_batch=48)
    try:
        out = torch.zeros((48, 64 * 96), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(48, dtype=torch.int32, device="cuda")
        h.detect_batch_device(frames.data_ptr(), 48, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
        h.batch_status()
        ref_c = cnt.cpu().numpy().copy()
        ref = np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(48, 64).copy()
        perm = torch.randperm(48, generator=torch.Generator().manual_seed(5)).cuda()
        staged = torch.zeros_like(frames)
        torch.cuda.synchronize()
        producer = torch.cuda.Stream()
        with torch.cuda.stream(producer):
            junk = frames.float()
            for _ in range(40):                              # keeps the producer busy for milliseconds
                junk = junk * 1.0001 + 0.5
            staged.copy_(frames[perm])
            ev = torch.cuda.Event()
            ev.record(producer)
        h.wait_event(ev.cuda_event)                          # no synchronise: the handle's stream waits for the producer
        h.detect_batch_device(staged.data_ptr(), 48, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
        h.batch_status()
        c = cnt.cpu().numpy()
        a = np.frombuffer(out.cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(48, 64)
        pi = perm.cpu().numpy()
        for j in range(48):
            assert c[j] == ref_c[pi[j]]
            assert a[j, :c[j]].tobytes() == ref[pi[j], :ref_c[pi[j]]].tobytes()
        assert int(c.sum()) > 900
        # the same through submit / wait
        h.set_pipeline_depth(2)
        staged.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(producer):
            junk = frames.float()
            for _ in range(40):
                junk = junk * 1.0001 + 0.5
            staged.copy_(frames[perm])
            ev2 = torch.cuda.Event()
            ev2.record(producer)
        h.wait_event(ev2.cuda_event)
        t = h.submit_device(staged.data_ptr(), 48, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
        h.wait(t)
        c2 = cnt.cpu().numpy()
        assert np.array_equal(c2, c)
    finally:
        h.close()


def test_batches_in_flight_equal_synchronous_batches(env):
    """arucohip_set_pipeline_depth / _submit / _wait: three batches on two pipeline lanes (host frames, host results) give
    the bytes of three synchronous arucohip_detect_batch calls; a third outstanding ticket is refused; the getters address
    the last batch waited for."""
    capi = env["capi"]
    frames, _ = env["synth"].make_stream(6, width=1920, height=1080, seed=41, device="cuda")
    fr = frames.cpu().numpy()
    K, dist = [1400, 0, 960, 0, 1400, 540, 0, 0, 1], [-0.1, 0.02, 1e-3, -5e-4, 0]
    h = capi.Handle(1920, 1080, max_batch=2)
    try:
        ref = [h.detect_batch_host(fr[2 * b:2 * b + 2], K=K, dist=dist, marker_size=0.05, cap=64) for b in range(3)]
        with pytest.raises(capi.ArucoHipError):          # no lanes yet
            h.submit_host(fr[0:2], np.zeros((2, 64), capi.MARKER_DTYPE), np.zeros(2, np.int32))
        h.set_pipeline_depth(2)
        outs = [np.zeros((2, 64), capi.MARKER_DTYPE) for _ in range(3)]
        ns = [np.zeros(2, np.int32) for _ in range(3)]
        batches = [np.ascontiguousarray(fr[2 * b:2 * b + 2]) for b in range(3)]
        t0 = h.submit_host(batches[0], outs[0], ns[0], K=K, dist=dist, marker_size=0.05)
        t1 = h.submit_host(batches[1], outs[1], ns[1], K=K, dist=dist, marker_size=0.05)
        with pytest.raises(capi.ArucoHipError) as e:     # both lanes busy
            h.submit_host(batches[2], outs[2], ns[2], K=K, dist=dist, marker_size=0.05)
        assert e.value.code == capi.E_CAPACITY
        h.wait(t0)
        t2 = h.submit_host(batches[2], outs[2], ns[2], K=K, dist=dist, marker_size=0.05)
        h.wait(t1)
        thr1 = h.thresholded(1, (1080, 1920))            # frame 1 of the batch of ticket t1 = frame 3 of the stream
        h.wait(t2)
        with pytest.raises(capi.ArucoHipError):          # a ticket is waited for once
            h.wait(t2)
        for b in range(3):
            for f in range(2):
                assert outs[b][f, :ns[b][f]].tobytes() == ref[b][f].tobytes()
        assert sum(int(n.sum()) for n in ns) >= 100
        h.set_pipeline_depth(0)
        h.detect_batch_host(fr[2:4])
        assert np.array_equal(h.thresholded(1, (1080, 1920)), thr1)
    finally:
        h.close()


def test_gl_modelviews_of_hip_detected_poses(env):
    """SURVEY §8 row f4 with poses that come from the HIP path (reference test Aruco.GL_Conversion, test/core_tests.cpp:
    230-283 <-> testdata/board/expected_gl.yml): detect the board image with intrinsics and marker size 1, board pose from
    BoardDetector, model-view matrices through the per-pose, the n-pose and the batched device entry points."""
    import json
    import os

    from tests.util import GOLDEN, rel_err
    capi = env["capi"]
    gray, board = load_case("board")
    gl = json.load(open(os.path.join(GOLDEN, "board_gl.json")))["gldata"]
    intr, bc = board["intrinsics"], board["board_conf"]
    h = capi.Handle(640, 480, max_batch=1)
    try:
        ms = h.detect(gray, K=intr["K"], dist=intr["dist"], marker_size=1.0)
        assert len(ms) == len(gl) - 2
        b = h.board_detect(ms, bc["ids"], bc["obj"], bc["info_type"], K=intr["K"], dist=intr["dist"], marker_size=1.0)
        assert rel_err(capi.gl_modelview(b["rvec"], b["tvec"]), gl[1]) < 1e-4          # north_star pose tolerance
        L = capi.load()
        mvn = np.zeros((len(ms), 16))
        assert L.arucohip_gl_modelview_n(ms.ctypes.data_as(C.c_void_p), len(ms), mvn.ctypes.data_as(C.c_void_p)) == 0
        batch = h.gl_modelview_batch(1)[0]
        assert batch.shape == (len(ms), 16)
        for i, m in enumerate(ms):
            assert rel_err(mvn[i], gl[2 + i]) < 1e-4, i
            assert np.array_equal(mvn[i], capi.gl_modelview(m["rvec"], m["tvec"]))
            assert np.max(np.abs(batch[i] - mvn[i])) < 1e-12                           # device sin / cos vs libm
        nopose = h.detect(gray)
        assert L.arucohip_gl_modelview_n(nopose.ctypes.data_as(C.c_void_p), len(nopose), mvn.ctypes.data_as(C.c_void_p)) == capi.E_INVALID
    finally:
        h.close()


def test_threshold_device_clock_span_and_event_interval_both_run(env):
    """arucohip_threshold_exec_ms (first wave in, last wave out by the device's constant-rate clock) and the hipEvent interval
    of the same launches, one batch at a time — the two ways bench.py times the dominant kernel: both count the four launches and
    give a positive time; the numbers are printed, not asserted (timing windows do not belong in the parity gate)."""
    capi, torch = env["capi"], env["torch"]
    from aruco_amd import synth
    n = 256
    fr, _ = synth.make_stream(n, seed=21, device="cuda")
    torch.cuda.synchronize()
    h = capi.Handle(1920, 1080, max_batch=n)
    try:
        out = torch.zeros((n, 64 * 96), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
        for _ in range(2):
            h.detect_batch_device(fr.data_ptr(), n, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
        h.batch_status()
        h.enable_timing(True)
        for _ in range(4):
            h.detect_batch_device(fr.data_ptr(), n, 1920, 1080, out.data_ptr(), 64, cnt.data_ptr())
        h.batch_status()
        ev = h.kernel_times()["threshold_kernel"]
        ms, launches = h.threshold_exec_ms()
        h.enable_timing(False)
        assert launches == 4
        clk = ms / launches
        # Both clocks ran and saw the same launches. How closely they agree (one batch at a time: 0.15 ms against 0.15-0.17 ms on a quiet
        # box; the event interval also holds the stamp reduction behind the kernel) is a measurement, not a parity condition: it is printed
        # (pytest -s / -rP) and kept out of the asserts, where a busy box would turn the whole -x parity run red (round 3: it did once).
        print("threshold launch: device clock %.4f ms, hipEvent interval %.4f ms" % (clk, ev))
        assert clk > 0 and ev > 0
    finally:
        h.close()


def test_gather_pipeline_on_the_device_with_rccl(env):
    """The N > 1 path of bench.py on the one GPU there is: a process group of ONE rank on backend nccl (= RCCL), the packed block of a real
    batch built by arucohip_compact_markers on the pipeline's own stream, the asynchronous gather on its own process group, the event the
    detector's stream waits for, three slots in flight — and what arrives equals the batch's result arrays. (Between distinct GPUs the same
    code has not run yet; the gloo tests cover world size 2.)"""
    import os
    import socket
    import torch.distributed as dist
    capi, torch = env["capi"], env["torch"]
    from aruco_amd import dist as adist
    if dist.is_initialized():
        pytest.skip("a process group exists already")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        B, CAPM, depth = 24, 64, 3
        frames, _ = env["synth"].make_stream(B, width=1920, height=1080, seed=29, device="cuda")
        h = capi.Handle(1920, 1080, max_batch=B)
        try:
            lib_stream = torch.cuda.Stream()
            h.set_stream(lib_stream.cuda_stream)
            h.set_pipeline_depth(depth)
            outs = [torch.zeros((B, CAPM * 96), dtype=torch.uint8, device="cuda") for _ in range(depth)]
            cnts = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(depth)]
            # capacity as bench.py agrees it
            h.detect_batch_device(frames.data_ptr(), B, 1920, 1080, outs[0].data_ptr(), CAPM, cnts[0].data_ptr())
            h.batch_status()
            cap_total = adist.agree_capacity(int(cnts[0].clamp(0, CAPM).sum().item()), B, CAPM, dev)
            gp = adist.GatherPipeline(B, CAPM, cap_total, depth, dev, always_collective=True)
            assert gp.bytes_per_step < 0.6 * (B * CAPM * 96)
            tickets = [None] * depth
            got = []
            for i in range(7):
                slot = i % depth
                if tickets[slot] is not None:
                    h.wait(tickets[slot])
                    ev = gp.submit(slot, outs[slot], cnts[slot])     # waits for the slot's previous gather first
                    lib_stream.wait_event(ev)
                tickets[slot] = h.submit_device(frames.data_ptr(), B, 1920, 1080, outs[slot].data_ptr(), CAPM, cnts[slot].data_ptr())
            for j in range(7 - depth, 7):
                slot = j % depth
                h.wait(tickets[slot])
                gp.submit(slot, outs[slot], cnts[slot])
            # the documented pattern of a rank-0 consumer: wait(slot), then read - WITHOUT drain() or a device synchronise in between.
            # wait() orders torch's current stream behind the gather, and unpack_block's copy to the host runs on that stream.
            last = (7 - 1) % depth
            c, fr, ovf = adist.unpack_block(gp.wait(last)[0], CAPM, capi.MARKER_DTYPE)
            ref_c = cnts[last].cpu().numpy()
            ref = np.frombuffer(outs[last].cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(B, CAPM)
            assert not ovf and np.array_equal(c, ref_c)
            for f in range(B):
                assert fr[f].tobytes() == ref[f, :ref_c[f]].tobytes()
            gp.drain()
            for slot in range(depth):
                blocks = gp.wait(slot)
                assert len(blocks) == 1
                c, fr, ovf = adist.unpack_block(blocks[0], CAPM, capi.MARKER_DTYPE)
                assert not ovf
                ref_c = cnts[slot].cpu().numpy()
                ref = np.frombuffer(outs[slot].cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(B, CAPM)
                assert np.array_equal(c, ref_c) and int(ref_c.sum()) > 400
                for f in range(B):
                    assert fr[f].tobytes() == ref[f, :ref_c[f]].tobytes()
        finally:
            h.close()
    finally:
        dist.destroy_process_group()
