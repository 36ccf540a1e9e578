#!/usr/bin/env python3
"""Headline benchmark: frames/s at 1920x1080 of the MarkerDetector::detect hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Both forms work: a plain `python bench.py --gpus N` (no RANK / WORLD_SIZE in the environment) starts the N ranks itself
through torch.distributed.run BEFORE anything in this process touches the GPU, waits for them and relays rank 0's JSON line.

One "step" = one arucohip_detect_batch over `--batch` synthetic 1080p frames (config 2 of BASELINE.json: ~20 markers
per frame, threshold + contours + decode + LINES refinement, no pose) that are already resident in HBM; results stay
in HBM. With N > 1 every rank owns its own camera stream (seed 4711 + rank, frames sharded one stream per GPU, no
data-path collective) and the per-frame marker blocks are gathered to rank 0 over RCCL once per step (config 5).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# d batches in flight use 2 d HIP streams (a main and a side stream per worker) next to the handle's own two and torch's. The HIP
# runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise. The variable must be in
# the environment before the runtime starts, so main() sets it right after parsing the arguments (INTEGRATION.md "deployment knobs").
# Measured: config 2 at depth 3 = 309.7k fps with 4 queues, 343.8k with 8, 320-331k with 16 (more kernels share the chip at once and
# all of them stretch); config 4 = 56.0k with 8 queues at depth 3 (two workers' border walks shared a queue), 73.8k with 12 or more,
# 94.4k with 16 queues and six batches in flight (its batches wait on one 5000-step border walk each).
# Round 3 (shorter kernels: the batches' dependent chains show again): the driver's 20-step run at depth 3 / 8 queues 463-465k, 4 / 12 471-480k, 5 / 12 483-487k,
# 6 / 16 470-477k (a deeper pipeline also fills and drains longer inside the timed region); 40 steps: 3 / 8 460-470k, 4 / 12 482k, 6 / 16 497k, 8 / 24 491k.
def default_hw_queues(config):
    return "16" if config == 4 else "12"

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 1920, 1080
ALG_BYTES_PER_FRAME = 3 * W * H          # SURVEY.md §8d's recipe: gray read + byte-image write + byte-image read (whole-pipeline figure)
HBM_PEAK_GBPS = 8000.0                   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
CAP = 64                                 # marker slots per frame in the gathered block (config 5)


def threshold_own_bytes(w, h):
    """ALGORITHMIC bytes per frame of the streaming kernel (threshold_eo_kernel) as it is built since round 2: it reads the gray frame once
    and writes the thresholded image as bits - the 8x8-pixel tiles (one zero pad tile row / column, aruco_amd/csrc/bits_tiles.h), the
    non-empty-tile bitmap (two words per 128-tile strip and tile row) and the four border lines of the lazy byte image (each padded to 16
    bytes, internal.h: thres_edge_stride). Nobody reads W*H back: the contour kernels read the tiles. 1080p: 2 073 600 + 262 208 + 4 352 +
    6 016 = 2 346 176 B = 1.131 W*H (SURVEY 8d's recipe charges 3 W*H, a byte image written and read again, which this pipeline replaced)."""
    tx, ty = max((w + 7) // 8 + 1, 4), max((h + 7) // 8 + 1, 4)
    strips = (w + 1023) // 1024
    return w * h + tx * ty * 8 + ty * 2 * strips * 8 + 2 * ((w + 15) & ~15) + 2 * ((h + 15) & ~15)


def h2d_ceiling_gbps(torch, dev, nbytes=1 << 30, reps=4):
    """Plain pinned-host -> device hipMemcpyAsync rate of THIS box (one stream, 1 GiB per copy): the ceiling the --host-frames leg is
    compared with."""
    src = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        dst.copy_(src, non_blocking=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        e1.record(st)
    e1.synchronize()
    return reps * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9


def thread_scaling_leg(capi, frame_1080p, threads=(1, 4, 16), seconds=1.5):
    """The reference's call shape under load: T host threads, each with a detector of its own (MarkerDetector is not re-entrant, one object
    per thread: src/markerdetector.cpp:334,372-380), each calling detect() on one pinned 1080p host frame per call in a loop
    (utils/aruco_test.cpp:153-160's frame loop, T cameras). frames/s over all threads; ctypes releases the GIL during the call."""
    import subprocess
    import tempfile
    import threading

    import numpy as np
    import torch

    exe = os.path.join(ROOT, "build", "thread_bench")
    if os.path.exists(exe):
        # the C++ host program (tools/thread_bench.cpp, built by __graft_entry__.build()): no interpreter between the threads and the C ABI
        with tempfile.NamedTemporaryFile(suffix=".raw", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tf:
            tf.write(np.ascontiguousarray(frame_1080p).tobytes())
            tf.flush()
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
            env["GPU_MAX_HW_QUEUES"] = "16"
            try:
                r = subprocess.run([exe, tf.name, str(frame_1080p.shape[1]), str(frame_1080p.shape[0]), str(seconds)] + [str(t) for t in threads],
                                   env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
                if r.returncode == 0:
                    d = json.loads(r.stdout.strip().splitlines()[-1])
                    d["host"] = "C++ (tools/thread_bench.cpp)"
                    return d
                sys.stderr.write("thread_bench failed (%d): %s\n" % (r.returncode, r.stderr[-300:]))
            except Exception as e:   # fall through to the Python threads
                sys.stderr.write("thread_bench: %r\n" % (e,))
    out = {"host": "python threads (ctypes releases the GIL during the call)"}
    for T in threads:
        pinned = [torch.empty(frame_1080p.shape, dtype=torch.uint8, pin_memory=True) for _ in range(T)]
        for p in pinned:
            p.copy_(torch.from_numpy(np.ascontiguousarray(frame_1080p)))
        frames = [p.numpy() for p in pinned]
        hs = [capi.Handle(frame_1080p.shape[1], frame_1080p.shape[0], max_batch=1) for _ in range(T)]
        counts, stop = [0] * T, threading.Event()
        try:
            for i in range(T):
                hs[i].detect(frames[i])

            def work(i):
                while not stop.is_set():
                    hs[i].detect(frames[i])
                    counts[i] += 1
            th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            time.sleep(seconds)
            stop.set()
            for t in th:
                t.join()
            out[str(T)] = round(sum(counts) / (time.perf_counter() - t0), 1)
        finally:
            for h in hs:
                h.close()
    return out


def cpu_baseline(frames_host, seconds_single=8.0, seconds_multi=12.0):
    """The CPU restatement of the reference algorithm (oracle/, kind "port") timed on this box's host cores on a
    bounded sample of the same frames: single thread, then one detector per core over frames (perf_tests.cpp style:
    gray input, wall-clock mean)."""
    import concurrent.futures as cf

    from oracle import orc

    n = len(frames_host)
    o = orc.Oracle()
    o.detect_raw(frames_host[0])
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds_single:
        o.detect_raw(frames_host[done % n])
        done += 1
    single = done / (time.perf_counter() - t0)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    oracles = [orc.Oracle() for _ in range(cores)]
    per_thread = max(2, int(single * seconds_multi))

    def work(i):
        k = 0
        for j in range(per_thread):
            oracles[i].detect_raw(frames_host[(i * per_thread + j) % n])
            k += 1
        return k

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(work, range(cores)))
    multi = total / (time.perf_counter() - t0)
    return {"value": round(multi, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d + %d detect() calls of the CPU restatement (oracle/) on %d of the bench's 1080p frames, "
                      "1 thread then %d threads over frames" % (done, total, n, cores),
            "single_thread_fps": round(single, 2)}


def latency_leg(capi, frame_1080p, runs=1000, cpu_seconds=1.5):
    """The reference's own performance tests (test/perf_tests.cpp: ArucoPerf.Single :31-56, .Board :58-86, .Multi :88-119,
    .HRM_Single :160-199, .GL_Conversion :121-158) on the same stills with the same settings: one gray frame in host memory per
    call, camera parameters given (poses are part of the call), BoardDetector::detect after it for Board / Multi, the d4x4_100
    dictionary and the test's detector settings for HRM_Single; mean wall-clock ms over `runs` calls, H2D of the frame and D2H of
    the results included. Both contour pipelines of the library (per-candidate walkers, waypoint segments: the default for single
    small frames) and the CPU restatement's ms beside them (1 thread). Plus one of the bench's 1080p frames (no pose)."""
    import numpy as np

    from aruco_amd.fixtures import load_case
    from oracle import orc

    out = {"method": "mean ms of %d calls on one host frame as in test/perf_tests.cpp (detect with camera parameters; + BoardDetector::detect for "
                     "board / chessboard; HRM settings for hrm), cpu = oracle restatement, 1 thread" % runs}
    cases = [("single", 1.0), ("board", 1.0), ("chessboard", 1.0), ("hrm", None), ("synthetic_1080p", None)]
    for name, msize in cases:
        if name == "synthetic_1080p":
            g, doc = np.ascontiguousarray(frame_1080p), {}
        else:
            g, doc = load_case(name)
        intr = doc.get("intrinsics")
        K, dist = (intr["K"], intr["dist"]) if intr else (None, None)
        st, dic, bc = doc.get("settings"), doc.get("dictionary"), doc.get("board_conf")
        if st:
            msize = st["marker_size"]
        if msize is None:
            msize = -1.0
        p = capi.default_params()
        okw = {}
        if st:   # core_tests.cpp:325-330 / perf_tests.cpp:172-177
            p.thres_param1, p.thres_param2, p.min_size, p.max_size, p.warp_size = st["thres_param1"], st["thres_param2"], st["min_size"], st["max_size"], st["warp_size"]
            okw = dict(thres_p1=st["thres_param1"], thres_p2=st["thres_param2"], min_size=st["min_size"], max_size=st["max_size"], warp_size=st["warp_size"])
        row = {}
        for mode in ("walkers", "segments"):
            os.environ["ARUCOHIP_CONTOURS"] = mode
            h = capi.Handle(g.shape[1], g.shape[0], max_batch=1, params=p)
            try:
                if dic:
                    h.set_dictionary(dic["markers"], dic["tau0"])

                def call():
                    m = h.detect(g, K=K, dist=dist, marker_size=msize)
                    if bc:
                        h.board_detect(m, bc["ids"], bc["obj"], bc["info_type"], K=K, dist=dist, marker_size=msize)
                    return m
                for _ in range(20):
                    got = call()
                t0 = time.perf_counter()
                for _ in range(runs):
                    call()
                row[mode + "_ms"] = round((time.perf_counter() - t0) / runs * 1e3, 4)
                row["markers"] = len(got)
            finally:
                h.close()
        os.environ.pop("ARUCOHIP_CONTOURS", None)
        o = orc.Oracle(**okw)
        if dic:
            o.set_hrm_dictionary(dic["markers"], dic["tau0"])

        def cpu_call():
            m = o.detect(g, K=K, dist=dist, marker_size=msize) if K is not None else o.detect_raw(g)
            if bc:
                orc.board_detect(m, bc["ids"], bc["obj"], bc["info_type"], K, dist, msize)
        cpu_call()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < cpu_seconds:
            cpu_call()
            n += 1
        row["cpu_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
        out[name] = row
    # ArucoPerf.GL_Conversion: projection matrix + the board's and every marker's model-view matrix from their poses (host arithmetic)
    g, doc = load_case("board")
    intr, bc = doc["intrinsics"], doc["board_conf"]
    h = capi.Handle(g.shape[1], g.shape[0], max_batch=1)
    try:
        m = h.detect(g, K=intr["K"], dist=intr["dist"], marker_size=1.0)
        b = h.board_detect(m, bc["ids"], bc["obj"], bc["info_type"], K=intr["K"], dist=intr["dist"], marker_size=1.0)
    finally:
        h.close()
    size = (g.shape[1], g.shape[0])
    t0 = time.perf_counter()
    for _ in range(runs):
        capi.gl_projection(intr["K"], size, size, 0.5, 10)
        capi.gl_modelview(b["rvec"], b["tvec"])
        for mk in m:
            capi.gl_modelview(mk["rvec"], mk["tvec"])
    out["gl_conversion"] = {"ms": round((time.perf_counter() - t0) / runs * 1e3, 4), "matrices": len(m) + 2,
                            "note": "ctypes call per matrix; the arithmetic is arucohip_gl_projection / arucohip_gl_modelview on the host"}
    return out


def launch_ranks(n):
    """Plain `python bench.py --gpus N`: this process has not touched the GPU (no torch import yet). Start N fresh rank
    processes with torch.distributed.run on 127.0.0.1, forward what they print (their one JSON line to stdout, everything
    else to stderr) and return their exit code. Nothing is re-exec'ed."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def stub_main(args, rank, world):
    """--stub: the launcher / rank plumbing without a GPU (CPU test of `--gpus N`): gloo, a fake step that produces marker
    blocks, the same barrier + max-over-ranks timing and the same overlapped, compacted gather (aruco_amd.dist.GatherPipeline) as the
    real run. Never a measurement."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from aruco_amd import dist as adist
    from aruco_amd.capi import MARKER_DTYPE

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    B, depth = min(args.batch, 8), 3
    blocks = np.zeros((B, CAP), MARKER_DTYPE)
    blocks["id"][:, 0] = rank
    mt = torch.from_numpy(blocks.view(np.uint8).reshape(B, CAP * 96).copy())
    ct = torch.ones(B, dtype=torch.int32)
    valid = B if (rank < world - 1 or world == 1) else B - B // 3      # ragged last shard: the last rank's stream ends early (padding frames, count 0)
    ct[valid:] = 0
    cap_total = adist.agree_capacity(int(ct.sum()), B, CAP, torch.device("cpu")) if world > 1 else B * CAP
    gp = adist.GatherPipeline(B, CAP, cap_total, depth, "cpu") if world > 1 else None
    last = None

    def step(i):
        time.sleep(0.002)
        if gp is not None:
            gp.submit(i % depth, mt, ct)       # waits for the gather this slot carried `depth` steps ago first

    for i in range(args.warmup):
        step(i)
    if gp is not None:
        gp.drain()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if gp is not None:
        last = gp.wait((args.steps - 1) % depth)
        gp.drain()
    if world > 1:
        dist.barrier()
    elapsed = adist.max_over_ranks(time.perf_counter() - t0, torch.device("cpu"))
    if rank == 0:
        ranks_seen, frames_seen = [0], valid
        if last is not None:
            un = [adist.unpack_block(blk, CAP, MARKER_DTYPE) for blk in last]
            ranks_seen = sorted(int(u[1][0][0]["id"]) for u in un)
            frames_seen = sum(int((u[0] > 0).sum()) for u in un)
        print(json.dumps({"metric": "frames/sec at %d×%d" % (W, H), "value": round(world * B * args.steps / elapsed, 2), "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
                          "data": "stub (launcher test, no GPU work)", "config": {"workload": "stub", "ranks_gathered": ranks_seen, "frames_gathered": frames_seen, "frames_per_rank": B,
                                                                                 "gather_bytes_per_rank_and_step": gp.bytes_per_step if gp else 0}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_leg(extra, timeout=420):
    """One of the other BASELINE configurations as a child process of its own (a fresh HIP runtime: config 4 wants 16 hardware queues),
    after this process has released its device memory. Returns the child's JSON line as a dict, or {"error": ...}."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--no-latency", "--no-legs"] + extra
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    if r.returncode != 0 or not lines:
        return {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=1024, help="frames per step and GPU")
    ap.add_argument("--frames", type=int, default=1024, help="distinct synthetic frames per GPU (SURVEY.md config 2: N=1024)")
    ap.add_argument("--pose", action="store_true", help="config 3: intrinsics + per-marker solvePnP")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4),
                    help="BASELINE.json config: 2 = 1080p stream no pose (headline), 3 = + per-marker solvePnP, "
                         "4 = 3840x2160 6x4 board frames + batched BoardDetector pose")
    ap.add_argument("--host-frames", action="store_true", help="frames start in pinned host memory (PCIe-inclusive rate)")
    ap.add_argument("--clutter", action="store_true", help="robustness leg: textured backgrounds behind the markers (not the headline)")
    ap.add_argument("--depth", type=int, default=0,
                    help="batches in flight (arucohip_detect_batch_submit / _wait); 1 = one synchronous-style batch at a time; default 5, "
                         "config 4: 6 (its 128-frame batches wait on a 5000-step border walk: more of them in flight fill the chip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency leg (extra keys of the JSON line)")
    ap.add_argument("--no-legs", action="store_true", help="skip the short legs of configs 3 / 4 / pinned H2D / clutter (extra key other_configs)")
    ap.add_argument("--profile-run", action="store_true",
                    help="for rocprofv3 (tools/profile.sh): warm-up + timed steps at the default depth only - no instrumented / isolated passes, no legs - "
                         "so that EVERY dispatch in the trace is an in-stream launch of the configuration the headline is measured on")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # launcher test on CPU (gloo), see stub_main
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", default_hw_queues(args.config))   # before torch / HIP start; the launched ranks inherit it

    in_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not args.stub:
        # fail before any GPU call, with a message, when the node has fewer devices than ranks were asked for (counting devices does not
        # initialise the HIP runtime on this image)
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) are visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); nothing was run\n" % (args.gpus, have))
            sys.exit(2)
    if args.gpus > 1 and not in_rank:
        sys.exit(launch_ranks(args.gpus))     # before torch / HIP are touched in this process
    if args.stub:
        return stub_main(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from aruco_amd import capi, synth
    from aruco_amd import dist as adist

    capi.load()
    build = capi.build_info()
    experiment = "ARUCOHIP_STAGE_EXPERIMENT" in build
    if args.config == 3:
        args.pose = True
    global W, H, ALG_BYTES_PER_FRAME
    board = None
    if args.config == 4:
        W, H = 3840, 2160
        ALG_BYTES_PER_FRAME = 3 * W * H
        if args.frames == 1024:
            args.frames = 128       # 4K frames are 4x larger; 128 distinct frames per GPU by default
        if args.batch == 1024:
            args.batch = 128
        from aruco_amd.fixtures import load_case
        _, doc = load_case("board")
        board = doc["board_conf"]
        K0 = np.array(doc["intrinsics"]["K"], np.float32).reshape(3, 3)   # CameraParameters::resize rule to 4K
        K0[0, 0] *= np.float32(W / 640.0); K0[0, 2] *= np.float32(W / 640.0)
        K0[1, 1] *= np.float32(H / 480.0); K0[1, 2] *= np.float32(H / 480.0)
        board["K"] = K0.reshape(-1)
    B = min(args.batch, args.frames)
    if board is None:
        frames, truth = synth.make_stream(args.frames, width=W, height=H, seed=4711 + rank, device=dev, clutter=args.clutter)
    else:
        frames, _ = synth.make_board_stream(args.frames, board["ids"], board["obj"], board["K"], width=W, height=H, seed=4711 + rank, device=dev)
        truth = [[{"id": i} for i in board["ids"]] for _ in range(args.frames)]
    torch.cuda.synchronize()
    frames_host = None
    if args.host_frames:
        frames_host = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True)
        frames_host.copy_(frames)
    limits = None
    if args.clutter:   # textured frames fill the walker lists an order of magnitude further than the flat stream
        limits = capi.Limits()
        capi.load().arucohip_default_limits(ctypes_byref(limits), W, H, B)
        limits.triggers_per_frame *= 4
        limits.long_walks_per_plane *= 4
        limits.contours_per_frame *= 2
    handle = capi.Handle(W, H, max_batch=B, device=local_rank, limits=limits)
    # the detector runs on a stream of its own: torch's current stream (and the stream RCCL orders itself against) is a different one,
    # so nothing the gather does can hold up a batch
    lib_stream = torch.cuda.Stream(device=dev)
    handle.set_stream(lib_stream.cuda_stream)
    depth = args.depth if args.depth > 0 else (6 if args.config == 4 else 5)
    outs = [torch.zeros((B, CAP * 96), dtype=torch.uint8, device=dev) for _ in range(depth)]
    cnts = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(depth)]
    out, cnt = outs[0], cnts[0]
    K = [1400, 0, 960, 0, 1400, 540, 0, 0, 1] if args.pose else None
    dcoef = [-0.10, 0.02, 1e-3, -5e-4, 0] if args.pose else None
    msize = 0.05 if args.pose else -1.0
    nwin = max(args.frames // B, 1)

    def src_of(off):
        return frames_host[off].data_ptr() if frames_host is not None else frames[off].data_ptr()   # host frames: H2D is part of the step

    # ---- the capacity of the packed result block from one pass over every window of the stream (N > 1: agreed between the ranks). The block
    # is what leaves the GPU every step: at N > 1 it is gathered to rank 0 over RCCL, at N = 1 it is copied to pinned host memory - both
    # asynchronously on the pipeline's own stream, waited for `depth` steps later, INSIDE the timed region (a consumer pays for it).
    gp, cap_total = None, B * CAP
    if not experiment:
        most = 0
        for w in range(nwin):
            if frames_host is not None:
                handle.detect_batch_mixed(src_of(w * B), B, W, H, out.data_ptr(), CAP, cnt.data_ptr(), K=K, dist=dcoef, marker_size=msize)
            else:
                handle.detect_batch_device(src_of(w * B), B, W, H, out.data_ptr(), CAP, cnt.data_ptr(), K=K, dist=dcoef, marker_size=msize)
            handle.batch_status()
            most = max(most, int(cnt.clamp(0, CAP).sum().item()))
        cap_total = adist.agree_capacity(most, B, CAP, dev)
        gp = adist.GatherPipeline(B, CAP, cap_total, depth, dev, to_host=(world == 1))
    if depth > 1:
        handle.set_pipeline_depth(depth)      # `depth` complete workers: batch i+1's threshold runs under batch i's border following
    tickets = [None] * depth
    slot_off = [0] * depth

    boards = []
    last = {"slot": 0}

    def finish(slot):
        """The batch that owns result slot `slot` is complete: board poses of its frames (config 4) and the gather (N > 1)."""
        if depth > 1:
            handle.wait(tickets[slot])          # raises on any device-side list overflow of that batch
            tickets[slot] = None
        last["slot"] = slot
        if board is not None:
            boards[:] = handle.board_detect_batch(B, board["ids"], board["obj"], board["info_type"], board["K"], [0.0] * 5, 0.039)
        if gp is not None:
            # pack + asynchronous gather on the pipeline's own stream and process group; it is waited for when this slot comes round
            # again, `depth` steps from now. The slot's next batch may overwrite the result arrays once the packing kernel has read them.
            ev = gp.submit(slot, outs[slot], cnts[slot])
            lib_stream.wait_event(ev)

    def step(i):
        off = (i % nwin) * B
        slot = i % depth
        if depth > 1:
            if tickets[slot] is not None:
                finish(slot)                    # the batch submitted `depth` steps ago frees its result arrays
            slot_off[slot] = off
            tickets[slot] = handle.submit_device(src_of(off), B, W, H, outs[slot].data_ptr(), CAP, cnts[slot].data_ptr(), K=K, dist=dcoef, marker_size=msize,
                                                 frames_on_device=frames_host is None)
            return
        slot_off[0] = off
        if frames_host is not None:
            handle.detect_batch_mixed(src_of(off), B, W, H, out.data_ptr(), CAP, cnt.data_ptr(), K=K, dist=dcoef, marker_size=msize)
        else:
            handle.detect_batch_device(src_of(off), B, W, H, out.data_ptr(), CAP, cnt.data_ptr(), K=K, dist=dcoef, marker_size=msize)
        handle.batch_status()                   # raises on any device-side list overflow
        finish(0)

    def drain(nsteps):
        """every batch still in flight completes (oldest first), and so does every gather"""
        if depth > 1:
            for j in range(max(0, nsteps - depth), nsteps):
                if tickets[j % depth] is not None:
                    finish(j % depth)
        if gp is not None:
            gp.drain()

    if args.profile_run:
        handle.enable_timing(2)                 # device-clock stamps of the threshold launches only (no events): the same launches rocprofv3 times
    for i in range(args.warmup):
        step(i)
    drain(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # ---- the timed region: K steps, no per-kernel instrumentation (no events, no clock stamps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain(args.steps)                           # all K batches and their gathers are complete inside the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = adist.max_over_ranks(elapsed, dev)

    # ---- correctness guard, right behind the timed region and on ITS last batch: the detected ids are rendered ids
    if experiment:
        # tools/stage_cost.sh: a variant built with -DARUCOHIP_STAGE_EXPERIMENT runs a truncated pipeline; only the step time means anything.
        # arucohip_build_info() of the loaded library says so - such a build never gets a headline line, whatever the environment says.
        print(json.dumps({"experiment": "stage_cost", "build": build, "stop_after": int(os.environ.get("ARUCOHIP_STOP_AFTER", "99")),
                          "ms_per_step": round(1e3 * elapsed / args.steps, 4), "invalid": "truncated pipeline, no results"}))
        return
    gslot = last["slot"]
    n_host = cnts[gslot].cpu().numpy()
    arr = np.frombuffer(outs[gslot].cpu().numpy().tobytes(), dtype=capi.MARKER_DTYPE).reshape(B, CAP)
    off = slot_off[gslot]
    found = 0
    for f in range(B):
        ids = set(int(x) for x in arr[f, :min(max(n_host[f], 0), CAP)]["id"])
        tids = set(t["id"] for t in truth[off + f])
        if not ids <= tids:
            raise SystemExit("frame %d: detected ids %s not a subset of rendered ids" % (f, sorted(ids - tids)))
        found += len(ids)
    rendered = sum(len(truth[off + f]) for f in range(B))
    if found < 0.9 * rendered:
        raise SystemExit("only %d of %d rendered markers detected" % (found, rendered))
    if board is not None and sum(b["has_pose"] for b in boards) < 0.95 * B:
        raise SystemExit("board pose missing on some frames")
    if gp is not None and world == 1:
        # N = 1: what arrived in pinned host memory for that batch equals its device result arrays
        c, fr, ovf = adist.unpack_block(gp.wait(gslot)[0], CAP, capi.MARKER_DTYPE)
        if ovf:
            raise SystemExit("the packed result block overflowed its capacity (%d markers)" % cap_total)
        for f in range(B):
            if fr[f] is None or fr[f].tobytes() != arr[f, :min(max(n_host[f], 0), CAP)].tobytes():
                raise SystemExit("host copy of the packed results differs from the device arrays at frame %d" % f)
    gathered = None
    if gp is not None and rank == 0 and world > 1:
        # what arrived on rank 0 for that batch: every rank's packed block, none overflowed, rank 0's own equals its result arrays
        gathered = gp.wait(gslot)
        tot = 0
        for r, blk in enumerate(gathered):
            c, fr, ovf = adist.unpack_block(blk, CAP, capi.MARKER_DTYPE)
            if ovf:
                raise SystemExit("rank %d: gather block overflowed its agreed capacity (%d markers)" % (r, cap_total))
            tot += int(np.clip(c, 0, CAP).sum())
            if r == 0:
                for f in range(B):
                    if fr[f] is None or fr[f].tobytes() != arr[f, :min(max(n_host[f], 0), CAP)].tobytes():
                        raise SystemExit("gathered block of rank 0 differs from its result arrays at frame %d" % f)
        if tot < 0.9 * rendered * world:
            raise SystemExit("gathered markers: %d of about %d" % (tot, rendered * world))

    if args.profile_run:
        pr_ms, pr_n = handle.threshold_exec_ms()
        handle.enable_timing(False)
        if rank == 0:
            print(json.dumps({"profile_run": True, "build": build, "threshold_device_clock_avg_ms": round(pr_ms / pr_n, 4) if pr_n else None, "threshold_launches_stamped": pr_n, "value": round(world * B * args.steps / elapsed, 2), "unit": "frames/s", "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "batches_in_flight": depth, "frames_per_launch": B,
                              "note": "every dispatch of this process is an in-stream launch at the bench's default depth (no instrumented passes)"}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    # ---- replica pass, outside the timed region: the timed loop once more (same steps, same batches in flight, same result copies) with nothing
    # but the threshold kernel's device-clock stamps on (no hipEvents between the kernels): the in-stream duration of its launches as
    # rocprofv3 --kernel-trace reports it for `bench.py --profile-run` (profiles/<tag>_kernel_stats.csv)
    handle.enable_timing(2)
    for i in range(args.steps):
        step(i)
    drain(args.steps)
    torch.cuda.synchronize()
    rep_ms, rep_n = handle.threshold_exec_ms()
    handle.enable_timing(False)
    # ---- instrumented passes, outside the timed region: per-kernel hipEvent intervals and the threshold kernel's device-clock span with
    # the same batches in flight, then one batch at a time (what rocprofv3 shows per dispatch when nothing else runs)
    gp_keep, gp = gp, None                      # the instrumented passes do not gather
    handle.enable_timing(True)
    nin = min(args.steps, 12)
    for i in range(nin):
        step(i)
    drain(nin)
    ktimes = handle.kernel_times()              # ms per launch, hipEvents on the launch streams
    exec_ms, exec_n = handle.threshold_exec_ms()  # the threshold launches by the device clock (first wave in, last wave out)
    chunks, per_launch = handle.batch_chunks()  # a step = `chunks` launches of every kernel, `per_launch` frames each
    handle.enable_timing(False)
    ktimes_iso = ktimes
    iso_exec_ms, iso_exec_n = exec_ms, exec_n
    if depth > 1:
        handle.enable_timing(True)
        for i in range(3):
            t = handle.submit_device(src_of(0), B, W, H, outs[0].data_ptr(), CAP, cnts[0].data_ptr(), K=K, dist=dcoef, marker_size=msize,
                                     frames_on_device=frames_host is None)
            handle.wait(t)
        ktimes_iso = handle.kernel_times()
        iso_exec_ms, iso_exec_n = handle.threshold_exec_ms()
        handle.enable_timing(False)
    fill = handle.debug_counters() if args.clutter else None

    if rank == 0:
        total_frames = world * B * args.steps
        fps = total_frames / elapsed
        # The roofline object is about the pipeline's one HBM-streaming kernel, the adaptive threshold pass. `achieved` / `frac` price it in
        # ITS OWN algorithmic bytes (threshold_own_bytes: gray read + bit tiles + bitmap + border lines written) over the duration of its
        # launches IN THE STREAM (batches in flight); SURVEY 8d's 3 W H recipe is kept as frac_survey_recipe, with the reason it can
        # exceed 1. The longest kernel of a batch is the border walk, a chain of dependent steps without a byte figure (`longest_kernel`).
        dom = "threshold_kernel"
        longest = max(ktimes_iso, key=lambda k: ktimes_iso[k])
        own = threshold_own_bytes(W, H)
        # in-stream duration: the device-clock span of each launch of the instrumented pass (first wave in to last wave out - what rocprofv3
        # reports per dispatch; tools/profile.sh --profile-run gives the same from the trace alone). The hipEvent interval (event_interval_ms)
        # also contains the time the dispatch queues behind the other batches' kernels.
        dom_ms = ktimes[dom]
        event_ms = dom_ms
        if rep_n > 0:
            dom_ms = rep_ms / rep_n
        elif exec_n > 0:
            dom_ms = exec_ms / exec_n
        achieved = own * per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        iso_ms = ktimes_iso.get(dom, dom_ms)
        # HBM bytes per launch: FETCH_SIZE + WRITE_SIZE of the rocprofv3 --pmc passes of THIS build at THIS depth (separate passes,
        # tools/profile.sh -> profiles/hbm_traffic.json), replayed - only when the file's build digest equals the loaded library's and the
        # passes were made at the depth this run uses; otherwise traffic is null and traffic_source says why (no stale replay).
        traffic, traffic_src, pipe_traffic, others = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if not os.path.exists(tpath):
            traffic_src = "no profiles/hbm_traffic.json"
        elif args.config not in (2, 3) or args.clutter or args.host_frames:
            traffic_src = "the committed PMC passes are of the default 1080p stream, not of this configuration"
        else:
            try:
                tj = json.load(open(tpath))
                if tj.get("build") != build:
                    traffic_src = "stale: profiles/hbm_traffic.json (%s) is of build '%s', the loaded library is '%s' - rerun tools/profile.sh" % (tj.get("tag", ""), tj.get("build"), build)
                elif tj.get("batches_in_flight") not in (None, depth):
                    traffic_src = "profiles/hbm_traffic.json (%s) was collected with %s batches in flight, this run uses %d" % (tj.get("tag", ""), tj.get("batches_in_flight"), depth)
                else:
                    kk = tj["kernels"]
                    traffic = kk[dom]["hbm_bytes_per_frame"] * per_launch
                    traffic_src = ("profiles/hbm_traffic.json (%s, same build, %s batches in flight): rocprofv3 --pmc FETCH_SIZE (x2 for this kernel's 16-byte-per-lane "
                                   "reads, MI355X_MICROARCH.md) + WRITE_SIZE, separate passes, replayed per launch" % (tj.get("tag", ""), tj.get("batches_in_flight")))
                    if args.config == 2:   # all kernels of a batch (the passes are of config 2)
                        pipe_traffic = sum(k["hbm_bytes_per_frame"] for k in kk.values()) * B
                        # traffic / algorithmic bytes of the three gather-type kernels (per frame): what each re-reads
                        tiles_b = max((W + 7) // 8 + 1, 4) * max((H + 7) // 8 + 1, 4) * 8
                        fillc = handle.debug_counters()
                        npts = fillc["points"] / max(B, 1)
                        alg = {"walker_long_kernel": (tiles_b, "the frame's bit-tile array once"),
                               "warp_hist_kernel": (48 * 56 * 56, "48 candidates x 56 x 56 nearest-neighbour samples"),
                               "contour_quad_kernel": (npts * 4 + tiles_b * 0.35, "4 B per kept contour point written + the tiles under the kept borders (~35 %% of the array)")}
                        others = {k: {"traffic_bytes_per_frame": round(kk[k]["hbm_bytes_per_frame"]), "algorithmic_bytes_per_frame": round(v[0]),
                                      "traffic_over_algorithmic": round(kk[k]["hbm_bytes_per_frame"] / v[0], 2), "algorithmic": v[1]} for k, v in alg.items() if k in kk}
            except Exception as e:   # a malformed file is reported, not replayed
                traffic, traffic_src = None, "profiles/hbm_traffic.json unreadable: %r" % (e,)
        res = {
            "metric": "frames/sec at %d×%d" % (W, H), "value": round(fps, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("3840x2160 6x4 board frames (24 markers), detect + batched BoardDetector solvePnP (config 4)" if board is not None else
                                    "1920x1080 synthetic stream, ~20 markers/frame, threshold+contour+decode+LINES"
                                    + (" + per-marker solvePnP (config 3)" if args.pose else ", no pose (config 2)"))
                                   + (", frames start in pinned host memory (PCIe inclusive)" if args.host_frames else "")
                                   + (", textured backgrounds (second headline: what camera frames look like)" if args.clutter else ""),
                       "frames_per_step_per_gpu": B, "distinct_frames_per_gpu": args.frames, "markers_rendered_per_frame": 24 if board is not None else 20,
                       "markers_detected_per_frame": round(found / B, 2), "batches_in_flight": depth, "build": build,
                       "timed_region": "no per-kernel instrumentation; every batch's packed result block (%d bytes) leaves the GPU inside it, asynchronously: "
                                       % (gp_keep.bytes_per_step if gp_keep is not None else 0)
                                       + ("copied to pinned host memory" if world == 1 else "gathered to rank 0 over RCCL") + ", waited for `batches_in_flight` steps later",
                       "thresholded_image": "written in the hot path" if os.environ.get("ARUCOHIP_THRES_BYTES", "0") not in ("", "0") else
                       "kept as bit tiles + border lines, bytes on request (ARUCOHIP_THRES_BYTES=1 writes them in the hot path)", "gpu_max_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "parallelism": "frames sharded 1 stream/GPU"
                       + (", packed marker blocks gathered over RCCL per step, asynchronously (own stream and process group), %d bytes per rank" % gp_keep.bytes_per_step
                          if (gp_keep is not None and world > 1) else "")},
            "hbm_algorithmic_gbps": round(ALG_BYTES_PER_FRAME * fps / 1e9, 2),
            "hbm_frac_of_peak": round(ALG_BYTES_PER_FRAME * fps / 1e9 / (HBM_PEAK_GBPS * world), 5),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": own * per_launch, "algorithmic_bytes_per_frame": own,
                         "algorithmic_bytes_formula": "W*H gray read + tiles_x*tiles_y*8 bit tiles + tiles_y*2*strips*8 bitmap + 2*Wp + 2*Hp border lines written "
                                                      "(bench.py: threshold_own_bytes; DESIGN.md 5)",
                         "frames_per_launch": per_launch, "launches_per_step": chunks, "avg_launch_ms": round(dom_ms, 4),
                         "avg_launch_ms_source": "device clock, first wave start to last wave end, %d launches of the replica pass (the timed loop again, %d batches in flight, "
                                                 "no hipEvents); profiles/: rocprofv3 average of the same kernel under bench.py --profile-run" % (rep_n, depth)
                         if rep_n > 0 else "device clock, launches of the hipEvent-instrumented pass" if exec_n > 0 else "hipEvent interval on the launch stream",
                         "avg_launch_ms_event_pass": round(exec_ms / exec_n, 4) if exec_n > 0 else None,
                         "traffic_over_algorithmic": round(traffic / (own * per_launch), 3) if traffic else None,
                         "frac_traffic": round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if traffic and dom_ms > 0 else None,
                         "event_interval_ms": round(event_ms, 4),
                         "isolated_launch_ms": round(iso_ms, 4),
                         "isolated_launch_ms_device_clock": round(iso_exec_ms / iso_exec_n, 4) if iso_exec_n > 0 else None,
                         "frac_isolated": round(own * per_launch / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if iso_ms > 0 else None,
                         "frac_traffic_isolated": round(traffic / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if traffic and iso_ms > 0 else None,
                         # SURVEY 8d's recipe: 3 W H per frame over the same in-stream duration. It charges a W*H byte-image write and a W*H
                         # byte-image read that this pipeline replaced by a bit image (1/8 of the bytes, written once), so it can exceed 1.
                         "frac_survey_recipe": round(ALG_BYTES_PER_FRAME * per_launch / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if dom_ms > 0 else None,
                         "frac_survey_recipe_note": "3*W*H per frame (SURVEY 8d) over avg_launch_ms; exceeds the kernel's real bytes 2.65x because the byte image it charges "
                                                    "(one write, one read) is kept as bits",
                         "frac_pipeline": round(ALG_BYTES_PER_FRAME * fps / 1e9 / (HBM_PEAK_GBPS * world), 5),
                         # every kernel's PMC bytes (replayed like `traffic`) over the step: how much of the peak the pipeline really moves
                         "pipeline_traffic_per_step": pipe_traffic,
                         "frac_pipeline_traffic": round(pipe_traffic / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS, 5) if pipe_traffic else None,
                         "pipeline_bound": "not HBM: vector-instruction issue and gather latency of the contour stages (DESIGN.md 6c/6d)",
                         "other_kernels": others,
                         "longest_kernel": longest, "longest_kernel_isolated_ms": round(ktimes_iso[longest], 4),
                         "longest_kernel_note": "a latency-bound chain of dependent border steps; SURVEY 8d defines no algorithmic bytes for it"},
            "kernel_ms_per_launch": {k: round(v, 4) for k, v in ktimes.items()},
            "kernel_ms_isolated": {k: round(v, 4) for k, v in ktimes_iso.items()},
        }
        if args.host_frames and world == 1:
            # the link: what this box's plain pinned hipMemcpyAsync reaches, and what the detector's frame copies reach inside the step
            ceil_gbps = h2d_ceiling_gbps(torch, dev)
            got_gbps = fps * W * H / 1e9
            res["h2d"] = {"plain_hipMemcpyAsync_gbps": round(ceil_gbps, 2), "achieved_gbps": round(got_gbps, 2), "frac_of_ceiling": round(got_gbps / ceil_gbps, 4),
                          "note": "pinned host -> device, 1 GiB copies on one stream, measured in this process after the timed region"}
        if fill is not None:
            res["list_fill_per_frame"] = {k: (round(v / B, 1) if k != "status" else v) for k, v in fill.items()}
        if world == 1 and not args.no_cpu_baseline:
            nsample = min(32, args.frames)
            res["cpu_baseline"] = cpu_baseline(frames[:nsample].cpu().numpy())
        frame0 = frames[0].cpu().numpy()
        legs = world == 1 and not args.no_legs and args.config == 2 and not args.host_frames and not args.clutter
        if (world == 1 and not args.no_latency and args.config != 4) or legs:
            handle.set_pipeline_depth(0) if depth > 1 else None      # free the lanes' memory before the small handles / the child processes
        if world == 1 and not args.no_latency and args.config != 4:
            res["latency"] = latency_leg(capi, frame0)
            res["latency"]["threads_x_detectors_1080p_fps"] = thread_scaling_leg(capi, frame0)
            res["latency"]["threads_note"] = ("T host threads, one detector each, one pinned 1080p host frame per detect() call (H2D, kernels, D2H per call): "
                                              "frames/s over all threads for T = 1, 4, 16")
        if legs:
            # the other configurations of BASELINE.json on the same clock as the headline: short runs, each a child process of its own
            handle.close()
            del frames, outs, cnts, out, cnt
            torch.cuda.empty_cache()
            oc = {}
            for key, extra in (("config3_fps", ["--config", "3", "--steps", "20", "--warmup", "5"]),    # like the headline: five batches in flight fill and drain inside the timed steps
                               ("config4_fps", ["--config", "4", "--steps", "30", "--warmup", "6"]),   # 128-frame batches, six in flight: 12 steps are two rounds
                               ("config2_pinned_h2d_fps", ["--host-frames", "--steps", "16", "--warmup", "5"]),   # 45 ms per step: five batches in flight fill and drain inside the timed steps
                               ("config2_clutter_fps", ["--clutter", "--steps", "20", "--warmup", "5"])):
                d = run_leg(extra)
                if "error" in d:
                    oc[key] = None
                    oc[key + "_error"] = d["error"]
                else:
                    oc[key] = d["value"]
                    oc[key.replace("_fps", "_ms_per_step")] = d["ms_per_step"]
                    oc[key.replace("_fps", "_steps")] = d["steps"]
                    if key == "config2_clutter_fps":
                        oc["config2_clutter_list_fill_per_frame"] = d.get("list_fill_per_frame")
                        oc["config2_clutter_kernel_ms_isolated"] = d.get("kernel_ms_isolated")
                    if key == "config2_pinned_h2d_fps":
                        oc["config2_pinned_h2d"] = d.get("h2d")
            oc["note"] = ("each a short run of `python bench.py` with the flags of that configuration in a child process (ids-subset guard as the "
                          "headline); config2_pinned_h2d = frames start in pinned host memory, H2D inside the step (h2d: the link rate it reaches against this box's plain "
                          "hipMemcpyAsync ceiling); clutter = the second headline: the same 1024-frame stream with textured backgrounds behind the markers")
            res["other_configs"] = oc
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def ctypes_byref(x):
    import ctypes
    return ctypes.byref(x)


if __name__ == "__main__":
    main()
