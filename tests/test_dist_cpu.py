"""Multi-process path on CPU: frames sharded round-robin over 2 ranks (gloo), marker blocks gathered to rank 0 and
re-interleaved — the N>1 data flow of bench.py / config 5 without a GPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aruco_amd import dist as adist
from aruco_amd.capi import MARKER_DTYPE

CAP = 8


def _fake_markers(frame):
    n = frame % 4 + 1
    m = np.zeros(CAP, MARKER_DTYPE)
    for i in range(n):
        m[i]["id"] = frame * 10 + i
        m[i]["corners"] = np.arange(8) + frame
        m[i]["tvec"] = [frame, i, 0.5]
    return m, n


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = adist.shard_indices(n_frames, rank, world)
    per_rank = (n_frames + world - 1) // world
    blocks = np.zeros((per_rank, CAP), MARKER_DTYPE)
    counts = np.zeros(per_rank, np.int32)
    for j, f in enumerate(mine):
        blocks[j], counts[j] = _fake_markers(f)
    mt = torch.from_numpy(blocks.view(np.uint8).reshape(per_rank, CAP * 96).copy())
    ct = torch.from_numpy(counts)
    ml, cl = adist.gather_marker_blocks(mt, ct, dst=0)
    t = adist.max_over_ranks(float(rank + 1), torch.device("cpu"))
    if rank == 0:
        out = adist.interleave_gathered(ml, cl, n_frames, CAP, MARKER_DTYPE)
        q.put((t, [o.tobytes() for o in out]))
    else:
        assert ml is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frames = 7   # ragged: rank 0 owns 4 frames, rank 1 owns 3 (padded)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    t, got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0
    for f in range(n_frames):
        m, n = _fake_markers(f)
        assert got[f] == m[:n].tobytes()


def _pipeline_worker(rank, world, port, n_frames, steps, depth, q):
    """Every rank runs `steps` batches; batch t of rank r holds the fake markers of frames {t * 1000 + f}. The overlapped, compacted
    gather (GatherPipeline, `depth` in flight) must deliver exactly what the blocking fixed-capacity gather delivers."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = adist.shard_indices(n_frames, rank, world)
    per_rank = (n_frames + world - 1) // world          # ragged: the last rank's last frame is padding (count 0)

    def batch(t):
        blocks = np.zeros((per_rank, CAP), MARKER_DTYPE)
        counts = np.zeros(per_rank, np.int32)
        for j, f in enumerate(mine):
            blocks[j], counts[j] = _fake_markers(t * 1000 + f)
        if t == 1 and rank == 1:
            counts[0] = -1                              # a frame that overflowed a device list travels as -1
        return torch.from_numpy(blocks.view(np.uint8).reshape(per_rank, CAP * 96).copy()), torch.from_numpy(counts)

    most = max(int(batch(t)[1].clamp(0, CAP).sum()) for t in range(steps))
    cap_total = adist.agree_capacity(most, per_rank, CAP, torch.device("cpu"))
    assert cap_total <= per_rank * CAP
    gp = adist.GatherPipeline(per_rank, CAP, cap_total, depth, "cpu")
    got, ref = {}, {}
    for t in range(steps):
        slot = t % depth
        if t >= depth:                                   # the slot's previous gather is waited for `depth` steps after it started
            blocks = gp.wait(slot)
            if rank == 0:
                got[t - depth] = [b.clone() for b in blocks]
        mt, ct = batch(t)
        gp.submit(slot, mt, ct)
        ml, cl = adist.gather_marker_blocks(mt, ct, dst=0)   # the blocking form, same batch
        if rank == 0:
            ref[t] = ([m.clone() for m in ml], [c.clone() for c in cl])
    for t in range(max(0, steps - depth), steps):
        blocks = gp.wait(t % depth)
        if rank == 0:
            got[t] = [b.clone() for b in blocks]
    gp.drain()
    if rank == 0:
        ok = True
        for t in range(steps):
            for r in range(world):
                counts, frames, ovf = adist.unpack_block(got[t][r], CAP, MARKER_DTYPE)
                m = np.frombuffer(ref[t][0][r].numpy().tobytes(), MARKER_DTYPE).reshape(per_rank, CAP)
                c = ref[t][1][r].numpy()
                ok &= not ovf and np.array_equal(counts, c)
                for j in range(per_rank):
                    if c[j] < 0:
                        ok &= frames[j] is None
                    else:
                        ok &= frames[j].tobytes() == m[j, :c[j]].tobytes()
        q.put((ok, gp.bytes_per_step, per_rank * CAP * 96 + per_rank * 4))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_compacted_gather_equals_the_blocking_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, 7, 8, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, packed, fixed = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    assert packed <= fixed + 32      # never more than the fixed-capacity arrays plus the 16-byte header (and padding)
    # at the bench's shape (1024 frames x 64 slots, ~20 markers per frame) the packed block is well under half of them
    assert adist.block_bytes(1024, 1024 * 26) < 0.45 * (1024 * 64 * 96 + 1024 * 4)


def test_pack_block_overflow_is_flagged():
    """A block packed for fewer marker slots than the batch holds: the counts are all there, the tail is missing and the flag says so."""
    blocks = np.zeros((3, CAP), MARKER_DTYPE)
    counts = np.array([3, 2, 4], np.int32)
    for f in range(3):
        blocks[f]["id"][:counts[f]] = 10 * f + np.arange(counts[f])
    mt = torch.from_numpy(blocks.view(np.uint8).reshape(3, CAP * 96).copy())
    blk = adist.pack_block(mt, torch.from_numpy(counts), CAP, 6)
    c, frames, ovf = adist.unpack_block(blk, CAP, MARKER_DTYPE)
    assert ovf and list(c) == [3, 2, 4]
    assert list(frames[0]["id"]) == [0, 1, 2] and list(frames[1]["id"]) == [10, 11] and list(frames[2]["id"]) == [20]
    full = adist.pack_block(mt, torch.from_numpy(counts), CAP, 9)
    c, frames, ovf = adist.unpack_block(full, CAP, MARKER_DTYPE)
    assert not ovf and list(frames[2]["id"]) == [20, 21, 22, 23]


def test_shard_indices_cover_everything():
    for n in (0, 1, 7, 8, 1024):
        for w in (1, 2, 4, 8):
            allidx = sorted(i for r in range(w) for i in adist.shard_indices(n, r, w))
            assert allidx == list(range(n))
            sizes = [len(adist.shard_indices(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a plain invocation (no RANK / WORLD_SIZE): bench.py starts the two ranks itself
    before touching the GPU and relays exactly one JSON line. Driven here with --stub (gloo, fake step) — the launcher,
    rendezvous on 127.0.0.1, gather and max-over-ranks timing are the real code."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["steps"] == 3 and doc["config"]["ranks_gathered"] == [0, 1]
    assert doc["value"] > 0 and doc["data"].startswith("stub")


def test_bench_stub_world_8_with_a_ragged_last_shard():
    """The launcher at the node's full width without a GPU: `python bench.py --gpus 8 --stub` starts eight gloo ranks on 127.0.0.1, every rank
    runs the overlapped packed gather (GatherPipeline, three in flight), the last rank's stream is ragged (a third of its frames are
    padding), and rank 0 ends up with one block of every rank and exactly the valid frames."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--stub"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    doc = json.loads(lines[0])
    B = doc["config"]["frames_per_rank"]
    assert doc["n_gpus"] == 8 and doc["config"]["ranks_gathered"] == list(range(8))
    assert doc["config"]["frames_gathered"] == 7 * B + (B - B // 3)
    assert doc["scaling"] == "weak" and doc["value"] > 0


def test_bench_refuses_more_ranks_than_devices():
    """`--gpus N` with fewer than N visible HIP devices: a clear message and a non-zero exit code BEFORE any GPU call or rank is started
    (this container has no GPU at all; on a one-GPU box the same holds for --gpus 2)."""
    import subprocess
    import sys

    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this box really has 8 devices")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 2
    assert "--gpus 8 but only" in r.stderr and r.stdout.strip() == ""
