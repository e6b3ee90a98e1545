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


class FrameGatherer:
    """The one data-path collective of an N-rank frame, with every buffer made ONCE: the padded send buffer, the per-rank
    receive buffers and the assembled frame on `dst`, the row indices of every rank.  gather(local_fb) then only copies
    (when this rank's row count is below the padded size), runs dist.gather (all_gather where the backend has no gather) and
    scatters the bands to their image rows — nothing is allocated inside a timed loop."""

    def __init__(self, height, width, band_rows=DEFAULT_BAND_ROWS, device="cpu", dtype=torch.float32, group=None, dst=0):
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.height, self.width, self.band_rows = height, width, band_rows
        self.counts = [len(shard_row_indices(height, band_rows, self.world, r)) for r in range(self.world)]
        self.pad_rows = max(self.counts)
        self.device = torch.device(device)
        self.send = torch.zeros((self.pad_rows, width, 3), dtype=dtype, device=self.device)
        self.parts = None
        self.frame = None
        self.index = None
        if self.rank == dst or _USE_ALL_GATHER:
            self._make_receive_side()

    def _make_receive_side(self):
        if self.parts is None:
            self.parts = [torch.empty_like(self.send) for _ in range(self.world)]
        if self.rank == self.dst and self.frame is None:
            self.frame = torch.empty((self.height, self.width, 3), dtype=self.send.dtype, device=self.device)
            self.index = [torch.as_tensor(shard_row_indices(self.height, self.band_rows, self.world, r), device=self.device)
                          for r in range(self.world)]

    def gather(self, local_fb):
        """local_fb: [local_rows, W, 3] tensor of this rank's rows → the full frame on `dst` (a buffer owned by this object,
        overwritten by the next call), None elsewhere."""
        global _USE_ALL_GATHER
        assert local_fb.shape[0] == self.counts[self.rank] and local_fb.shape[1] == self.width, (local_fb.shape, self.counts, self.rank)
        if self.world == 1:
            return local_fb
        if local_fb.shape[0] == self.pad_rows and local_fb.is_contiguous() and local_fb.device == self.device:
            send = local_fb
        else:
            self.send[:local_fb.shape[0]].copy_(local_fb)
            send = self.send
        if not _USE_ALL_GATHER:
            try:
                dist.gather(send, self.parts if self.rank == self.dst else None, dst=self.dst, group=self.group)
            except (RuntimeError, NotImplementedError):
                # a backend without gather: every rank takes this branch on the same call (the error is
                # raised before any communication), then all of them use all_gather from here on
                _USE_ALL_GATHER = True
        if _USE_ALL_GATHER:
            self._make_receive_side()
            dist.all_gather(self.parts, send, group=self.group)
        if self.rank != self.dst:
            return None
        for r in range(self.world):
            self.frame.index_copy_(0, self.index[r], self.parts[r][:self.counts[r]])
        return self.frame


def gather_frame(local_fb, height, band_rows, group=None, dst=0):
    """One-off form of FrameGatherer.gather (allocates its buffers per call): local_fb [local_rows, W, 3] → the full
    [height, W, 3] frame on `dst`, None elsewhere."""
    if dist.get_world_size(group) == 1:
        return local_fb
    g = FrameGatherer(height, local_fb.shape[1], band_rows, device=local_fb.device, dtype=local_fb.dtype, group=group, dst=dst)
    return g.gather(local_fb)
