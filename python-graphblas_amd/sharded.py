"""1-D row sharding of mxv over the GPUs of one node: rank r owns the row block
[lo, hi) of A and a full replica of u; after the local product the ranks all-gather their w slices into
the next u (RCCL over xGMI on GPUs -- torch.distributed backend "nccl"; "gloo" in the CPU tests).

Blocks are equal and 64-row aligned, so the all-gather lands directly in the replicated vector's HBM
image (values AND bit-packed presence words), with no staging copy."""
from __future__ import annotations


def row_block(n, rank, world):
    if n % (64 * world):
        raise ValueError("n must be a multiple of 64 * world_size for zero-copy all-gather")
    rows = n // world
    return rank * rows, (rank + 1) * rows


def allgather_into(u_full, w_local, *, device="cuda", values=True, presence=True):
    """u_full[lo_r:hi_r] = w_local of rank r, for every r (one collective per array)."""
    import torch.distributed as dist

    from . import device as dev

    u_vals, u_words = dev.vector_device_views(u_full, device)
    w_vals, w_words = dev.vector_device_views(w_local, device)
    if values:
        _gather(dist, u_vals, w_vals)
    if presence:
        _gather(dist, u_words, w_words)
        dev.vector_modified(u_full)


def _gather(dist, out, part):
    try:
        dist.all_gather_into_tensor(out, part)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant
        parts = list(out.chunk(dist.get_world_size()))
        dist.all_gather(parts, part)
