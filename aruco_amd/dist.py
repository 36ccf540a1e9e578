"""Frame-sharded multi-GPU driver: frames are independent units, frame f (or camera c) goes to rank
f mod world; there is no collective on the data path. The only exchange is the gather of the per-frame marker blocks
({int32 n, arucohip_marker_t[cap]}) to rank 0 once per batch — RCCL over xGMI with backend "nccl", "gloo" on CPU tests.

Round 3: the gather is off the step loop's critical path and compact.
  * `pack_block` / `arucohip_compact_markers` turn a batch's fixed-capacity arrays ([B][cap] markers + [B] counts; 6.3 MB per rank at the
    bench's config 2, a third of it used) into one contiguous block {total, nframes, cap_total, overflow, counts[B], markers[cap_total]}
    with the frames' markers back to back; `cap_total` is agreed between the ranks once (`agree_capacity`), so every rank sends the same
    number of bytes and ONE fixed-size gather per batch carries counts and markers together.
  * `GatherPipeline` issues that gather asynchronously (`async_op=True`) on a process group and a stream of its own, ordered behind the
    batch's completion, and waits for it `depth` steps later: the detector's stream never waits for RCCL and the next batches' kernels run
    under the transfer. Results equal the blocking `gather_marker_blocks` (tests/test_dist_cpu.py).
"""
import numpy as np
import torch
import torch.distributed as dist

MARKER_BYTES = 96


def shard_indices(n_units, rank, world):
    """Units (frames / cameras) owned by `rank`: round-robin, so every rank gets floor or ceil of n/world."""
    return list(range(rank, n_units, world))


def gather_marker_blocks(markers_u8, counts_i32, dst=0, group=None):
    """Blocking gather of the fixed-capacity marker blocks from every rank to `dst` (the round-1/2 form; kept as the reference the
    overlapped gather is tested against).

    markers_u8: uint8 tensor [frames_local, cap*96]; counts_i32: int32 tensor [frames_local]. All ranks must pass the
    same shapes (pad the last batch). Returns (list of per-rank marker tensors, list of per-rank count tensors) on
    `dst`, (None, None) elsewhere. Works with nccl (device tensors) and gloo (CPU tensors)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [markers_u8], [counts_i32]
    if rank == dst:
        mlist = [torch.empty_like(markers_u8) for _ in range(world)]
        clist = [torch.empty_like(counts_i32) for _ in range(world)]
    else:
        mlist = clist = None
    dist.gather(markers_u8, mlist, dst=dst, group=group)
    dist.gather(counts_i32, clist, dst=dst, group=group)
    return mlist, clist


def interleave_gathered(mlist, clist, n_units, cap, marker_dtype):
    """Undo the round-robin sharding on rank 0: returns a list (length n_units) of numpy structured marker arrays."""
    world = len(mlist)
    out = [None] * n_units
    for r in range(world):
        m = np.frombuffer(mlist[r].cpu().numpy().tobytes(), dtype=marker_dtype).reshape(-1, cap)
        c = clist[r].cpu().numpy()
        for j, f in enumerate(shard_indices(n_units, r, world)):
            out[f] = m[j, :min(int(c[j]), cap)].copy()
    return out


def max_over_ranks(value, device):
    """Max of a python float over all ranks (the bench contract's timing rule)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- the packed gather block (layout of arucohip_compact_markers, include/arucohip.h)
def block_head_bytes(nframes):
    return (16 + 4 * nframes + 15) & ~15


def block_bytes(nframes, cap_total):
    return block_head_bytes(nframes) + cap_total * MARKER_BYTES


def pack_block(markers_u8, counts_i32, cap, cap_total):
    """Host statement of arucohip_compact_markers on CPU tensors (the gloo tests and the checker of the device kernel): returns the
    packed uint8 tensor."""
    nf = int(counts_i32.numel())
    m = markers_u8.contiguous().view(torch.uint8).reshape(nf, cap, MARKER_BYTES).numpy()
    c = counts_i32.numpy().astype(np.int64)
    clip = np.clip(c, 0, cap)
    out = np.zeros(block_bytes(nf, cap_total), np.uint8)
    head = out[:block_head_bytes(nf)].view(np.int32)
    head[4:4 + nf] = counts_i32.numpy()
    body = out[block_head_bytes(nf):].reshape(cap_total, MARKER_BYTES)
    at = 0
    for f in range(nf):
        k = int(max(0, min(clip[f], cap_total - at)))
        if k:
            body[at:at + k] = m[f, :k]
        at += int(clip[f])
    head[0], head[1], head[2], head[3] = at, nf, cap_total, 1 if at > cap_total else 0
    return torch.from_numpy(out)


def unpack_block(block_u8, cap, marker_dtype):
    """Packed block (uint8 tensor, any device) -> (counts int32[nframes], list of per-frame structured marker arrays, overflow flag).
    Frames whose count is -1 (a device list overflowed) come back as None."""
    raw = block_u8.detach().cpu().numpy()
    head = raw[:16].view(np.int32)
    total, nf, cap_total, overflow = (int(v) for v in head)
    counts = raw[16:16 + 4 * nf].view(np.int32).copy()
    body = raw[block_head_bytes(nf):block_head_bytes(nf) + cap_total * MARKER_BYTES].view(marker_dtype)
    frames, at = [], 0
    for f in range(nf):
        k = int(min(max(int(counts[f]), 0), cap))
        if counts[f] < 0:
            frames.append(None)
        else:
            frames.append(body[at:min(at + k, cap_total)].copy())
        at += k
    assert at == total
    return counts, frames, bool(overflow)


def agree_capacity(local_total, nframes, cap, device, headroom=1.25, group=None):
    """Marker slots of the packed block, the same on every rank: the largest per-batch total any rank has seen (`local_total`) with
    headroom, at most nframes * cap (what the fixed-capacity arrays hold — always enough)."""
    t = torch.tensor([int(local_total)], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    want = int(int(t.item()) * headroom) + 64
    return int(min(nframes * cap, (want + 255) & ~255))


class GatherPipeline:
    """Overlapped gather of packed marker blocks to rank `dst`, one per batch, `depth` batches in flight.

    submit(slot, markers, counts): called when the batch that owns result slot `slot` is COMPLETE (after the detector's wait). Packs
    the slot's arrays into the slot's send block — on the GPU with arucohip_compact_markers on this object's own stream, on the CPU with
    pack_block — and starts the asynchronous gather. Returns the event (GPU) after which the slot's result arrays may be overwritten.
    wait(slot): the gather of that slot is complete; on `dst` returns the list of per-rank packed blocks (views of the receive buffer,
    valid until the slot's next submit), elsewhere None.
    The collective runs on a process group of its own, so it never queues behind (or in front of) the caller's barriers and reductions."""

    def __init__(self, nframes, cap, cap_total, depth, device, dst=0, new_group=True, always_collective=False, to_host=False):
        self.nframes, self.cap, self.cap_total, self.depth, self.dst = nframes, cap, cap_total, depth, dst
        self.device = torch.device(device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        # always_collective: issue the gather also in a world of one rank (the single-GPU test of the RCCL code path)
        self.collective = self.world > 1 or always_collective
        self.group = dist.new_group() if (new_group and self.collective) else None
        self.on_gpu = self.device.type == "cuda"
        nbytes = block_bytes(nframes, cap_total)
        self.send = [torch.zeros(nbytes, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.recv = [[torch.zeros(nbytes, dtype=torch.uint8, device=self.device) for _ in range(self.world)] if self.rank == dst else None
                     for _ in range(depth)]
        self.work = [None] * depth
        self.stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.bytes_per_step = nbytes
        # to_host (one rank, no collective): the packed block of every batch is also copied to pinned host memory on this object's stream -
        # what a single-GPU consumer of the results pays per batch; wait(slot) then returns the host block
        self.to_host = bool(to_host) and self.on_gpu and not self.collective
        self.host = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(depth)] if self.to_host else None
        self.host_ev = [None] * depth

    def submit(self, slot, markers_u8, counts_i32):
        if self.work[slot] is not None or self.host_ev[slot] is not None:
            self.wait(slot)
        ev = None
        if self.on_gpu:
            from . import capi
            with torch.cuda.stream(self.stream):
                capi.compact_markers(markers_u8.data_ptr(), counts_i32.data_ptr(), self.nframes, self.cap, self.send[slot].data_ptr(), self.cap_total,
                                     self.stream.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                if self.collective:
                    self.work[slot] = dist.gather(self.send[slot], self.recv[slot], dst=self.dst, group=self.group, async_op=True)
                elif self.to_host:
                    self.host[slot].copy_(self.send[slot], non_blocking=True)       # D2H behind the packing kernel, on this object's stream
                    self.host_ev[slot] = torch.cuda.Event()
                    self.host_ev[slot].record(self.stream)
        else:
            self.send[slot].copy_(pack_block(markers_u8, counts_i32, self.cap, self.cap_total))
            if self.collective:
                self.work[slot] = dist.gather(self.send[slot], self.recv[slot], dst=self.dst, group=self.group, async_op=True)
        return ev

    def wait(self, slot):
        w = self.work[slot]
        if w is not None:
            if self.on_gpu:
                with torch.cuda.stream(self.stream):
                    w.wait()          # this object's stream waits for RCCL; the detector's stream is not involved
                # ... and the CONSUMER is ordered behind it: for RCCL w.wait() only makes self.stream wait, the host does not block. Whoever reads
                # the returned blocks does so on torch's current stream (unpack_block's .cpu() is a copy on it, and the host waits for that
                # copy), so that stream now waits for this object's stream and therefore for the gather.
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
            else:
                w.wait()
            self.work[slot] = None
        if self.to_host:
            if self.host_ev[slot] is not None:
                self.host_ev[slot].synchronize()      # the host blocks until the copy of that batch's block has landed
                self.host_ev[slot] = None
            return [self.host[slot]]
        if not self.collective:
            return [self.send[slot]]
        return self.recv[slot] if self.rank == self.dst else None

    def drain(self):
        for s in range(self.depth):
            self.wait(s)
        if self.on_gpu:
            self.stream.synchronize()
