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
