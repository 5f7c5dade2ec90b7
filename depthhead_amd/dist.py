"""Multi-GPU plumbing: frames are independent units, so they shard across ranks with no
data-path collective; the only exchange is the gather of the 40-byte pose records at the end of a
batch.  One process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for the tests).

The gather moves n_frames x 40 B (10 KB for 256 frames per rank): latency-bound, so it is ONE
all-gather per batch, never a hand-rolled ring.
"""
from __future__ import annotations

import numpy as np

from ._lib import POSE_DTYPE

POSE_BYTES = POSE_DTYPE.itemsize


def shard_range(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """Rank r of R takes frames [r*N/R, (r+1)*N/R) (integer arithmetic: contiguous, disjoint,
    covering, sizes differ by at most one)."""
    if world <= 0 or not 0 <= rank < world or n_frames < 0:
        raise ValueError("bad shard arguments")
    return (rank * n_frames) // world, ((rank + 1) * n_frames) // world


def gather_poses(local_bytes, n_total: int, group=None):
    """All-gather per-rank pose buffers (uint8 tensors of n_local*40 bytes, device or CPU) into one
    uint8 tensor of n_total*40 bytes ordered by rank.  Ranks may hold different frame counts
    (ragged shards): buffers are padded to the largest shard for the collective and trimmed after.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    counts = [(b - a) * POSE_BYTES for a, b in sizes]
    if local_bytes.numel() != counts[rank]:
        raise ValueError(f"rank {rank}: expected {counts[rank]} pose bytes, got {local_bytes.numel()}")
    mx = max(counts)
    if mx == 0:
        return torch.zeros(0, dtype=torch.uint8, device=local_bytes.device)
    send = local_bytes
    if send.numel() != mx:
        send = torch.zeros(mx, dtype=torch.uint8, device=local_bytes.device)
        send[: local_bytes.numel()] = local_bytes
    recv = torch.empty(world * mx, dtype=torch.uint8, device=local_bytes.device)
    try:
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):   # older gloo: fall back to the list form
        parts = [torch.empty(mx, dtype=torch.uint8, device=local_bytes.device) for _ in range(world)]
        dist.all_gather(parts, send.contiguous(), group=group)
        recv = torch.cat(parts)
    if all(c == mx for c in counts):
        return recv
    return torch.cat([recv[r * mx: r * mx + counts[r]] for r in range(world)])


def poses_from_bytes(buf) -> np.ndarray:
    """uint8 tensor / array -> POSE_DTYPE structured array."""
    if hasattr(buf, "cpu"):
        buf = buf.cpu().numpy()
    return np.frombuffer(np.ascontiguousarray(buf).tobytes(), dtype=POSE_DTYPE)
