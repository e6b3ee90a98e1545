"""Multi-GPU frame assembly: one process per GPU, image rows sharded in interleaved bands,
one gather to rank 0 at frame end.

Pixels are independent and the RNG is a pure function of (column, row, sample)
(reference src/camera.cu:25-28), so every rank renders a disjoint set of rows of the SAME frame
with the scene replicated; the assembled frame is bit-identical to a single-GPU render.  Bands are
interleaved (band b → rank b % world) so sky and geometry rows are spread over the ranks.
The only data-path collective is the gather of the row bands (RCCL over xGMI on GPUs, gloo on CPU).
"""
import numpy as np
import torch
import torch.distributed as dist

import rtp_bindings as rb

DEFAULT_BAND_ROWS = 8
_USE_ALL_GATHER = False


def shard_for_rank(rank, world_size, band_rows=DEFAULT_BAND_ROWS):
    return rb.Shard(band_rows, world_size, rank)


def shard_row_indices(height, band_rows, world_size, rank):
    """Image rows of `rank`, in the order rt_render writes them (increasing)."""
    rows = np.arange(height)
    return rows[(rows // band_rows) % world_size == rank]


def gather_frame(local_fb, height, band_rows, group=None, dst=0):
    """local_fb: [local_rows, W, 3] float32 tensor of this rank's rows.  Returns the full
    [height, W, 3] frame on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    width = local_fb.shape[1]
    counts = [len(shard_row_indices(height, band_rows, world, r)) for r in range(world)]
    assert local_fb.shape[0] == counts[rank], (local_fb.shape, counts, rank)
    if world == 1:
        return local_fb
    pad_rows = max(counts)
    send = local_fb
    if send.shape[0] != pad_rows:
        send = torch.zeros((pad_rows, width, 3), dtype=local_fb.dtype, device=local_fb.device)
        send[:local_fb.shape[0]] = local_fb
    send = send.contiguous()
    global _USE_ALL_GATHER
    parts = None
    if not _USE_ALL_GATHER:
        try:
            if rank == dst:
                parts = [torch.empty_like(send) for _ in range(world)]
                dist.gather(send, parts, dst=dst, group=group)
            else:
                dist.gather(send, None, dst=dst, group=group)
        except (RuntimeError, NotImplementedError):
            # a backend without gather: every rank takes this branch on the same call (the error is
            # raised before any communication), then all of them use all_gather from here on
            _USE_ALL_GATHER = True
            parts = None
    if _USE_ALL_GATHER:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
    if rank != dst:
        return None
    frame = torch.empty((height, width, 3), dtype=local_fb.dtype, device=local_fb.device)
    for r in range(world):
        idx = torch.as_tensor(shard_row_indices(height, band_rows, world, r), device=local_fb.device)
        frame.index_copy_(0, idx, parts[r][:counts[r]])
    return frame
