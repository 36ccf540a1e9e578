"""Frame-sharded multi-GPU driver: frames are independent units, frame f (or camera c) goes to rank
f mod world; there is no collective on the data path. The only exchange is the gather of the per-frame marker blocks
({int32 n, arucohip_marker_t[cap]}) to rank 0 once per batch — RCCL over xGMI with backend "nccl", "gloo" on CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

MARKER_BYTES = 96


def shard_indices(n_units, rank, world):
    """Units (frames / cameras) owned by `rank`: round-robin, so every rank gets floor or ceil of n/world."""
    return list(range(rank, n_units, world))


def gather_marker_blocks(markers_u8, counts_i32, dst=0, group=None):
    """Gather fixed-capacity marker blocks from every rank to `dst`.

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
