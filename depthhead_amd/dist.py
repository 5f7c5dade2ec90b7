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


def predict_stream(hp, frames, intrinsic, group=None) -> np.ndarray:
    """Frame-sharded `predict_parameter_parallel` over a stream every rank can see (a host array or a
    memory-mapped file of [N, H, W] uint16 frames): rank r predicts frames [r*N/R, (r+1)*N/R) with its own
    predictor `hp` (any object with `predict_batch(frames, intrinsic)` returning POSE_DTYPE records -- a
    `HoughPrediction` on this rank's GPU), then ONE all-gather of the 40-byte pose records hands every rank
    the poses of all N frames in stream order.  No data-path collective."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return hp.predict_batch(np.asarray(frames), intrinsic)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_total = len(frames)
    a, b = shard_range(n_total, rank, world)
    local = hp.predict_batch(np.ascontiguousarray(frames[a:b]), intrinsic) if b > a else np.zeros(0, dtype=POSE_DTYPE)
    buf = torch.from_numpy(np.frombuffer(local.tobytes(), dtype=np.uint8).copy())
    backend = dist.get_backend(group)
    if backend == "nccl":                      # RCCL moves device buffers
        buf = buf.to(torch.device("cuda", torch.cuda.current_device()))
    return poses_from_bytes(gather_poses(buf, n_total, group))


class ShardedPredictor:
    """The steady-state form of `predict_stream` for device-resident shards (what `bench.py --gpus N` times):
    this rank's `n_local` frames already sit in HBM; `submit()` enqueues the local batch and, on the same stream behind its
    kernels, the all-gather of its pose records; nothing waits on the host: buffers alternate so that the gather of
    step i overlaps the kernels of step i + 1 (other streams), and `fence()` drains everything.

    `hp` may be ONE predictor or a list of them (pipeline depth = len): consecutive steps then alternate between the
    predictors, each on a stream of its own, so that the latency-bound tail kernels of batch i (vote, mean shift) run
    beside the bandwidth / issue-bound head kernels of batch i + 1 (`dh_predictor` objects are independent: "distinct
    predictors may run concurrently", include/depthhead_hip.h).  Measured on one MI355X (round 2, final kernels): 513 k frames/s at depth 1,
    591-613 k at depth 2, 627-657 k at depth 4, no more beyond (`tools/experiments/pipeline_sweep.sh`).  Every step still processes one whole batch through every kernel.

    With `backend == "gloo"` (rehearsal on one device, or CPU-only hosts) the records are staged through host memory.
    World size 1: no collective at all -- unless `force_collective` (a process group of ONE rank still runs RCCL's
    all-gather on device buffers: that is how the collective path is executed on a one-GPU box, `bench.py --force-dist`)."""

    def __init__(self, hp, n_local: int, w: int, h: int, intrinsic, group=None, device=None, force_collective: bool = False):
        import torch
        import torch.distributed as dist
        self.hps = list(hp) if isinstance(hp, (list, tuple)) else [hp]
        self.hp = self.hps[0]
        self.depth = len(self.hps)
        self.n_local, self.w, self.h, self.intrinsic, self.group = n_local, w, h, intrinsic, group
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.collective = self.dist is not None and (self.world > 1 or force_collective)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.on_device = self.dist is None or self.dist.get_backend(group) == "nccl"
        nb = n_local * POSE_BYTES
        # one pose / gather buffer pair per slot; with a single predictor two slots still alternate (gather i beside kernels i + 1)
        self.slots = max(2, self.depth) if self.collective else self.depth
        self.pose_bufs = [torch.zeros(nb, dtype=torch.uint8, device=self.device) for _ in range(self.slots)]
        gdev = self.device if self.on_device else torch.device("cpu")
        self.gathered = [torch.zeros(self.world * nb, dtype=torch.uint8, device=gdev) for _ in range(self.slots)] if self.collective else None
        self.pending = [None] * self.slots
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.depth)] if self.depth > 1 else None
        if self.depth > 1:
            # several predictors in flight ARE the overlap that the library's forked sub-batches (calls of >= 512 frames) would
            # add, without the fork's event plumbing: 512 frames per call, four in flight, 669 k frames/s forked, 713 k whole
            for q in self.hps:
                q.set_forking(1)
        self.count = 0
        self.graph = False

    def capture(self, frames_ptr: int):
        """Replay the local batch from a hipGraph (launch-bound configs).  One predictor and one pose buffer then: a
        replay writes where the capture wrote, so each step waits for its own gather before the next replay."""
        self.hp.graph_capture(frames_ptr, self.n_local, self.w, self.h, self.intrinsic, self.pose_bufs[0].data_ptr())
        self.graph = True
        self.depth, self.hps, self.streams = 1, self.hps[:1], None

    def submit(self, frames_ptr: int, stream) -> int:
        """Enqueue one step; returns the slot.  At depth 1 the step runs on `stream`; deeper pipelines bring their own
        streams, each of which first waits for the work already enqueued on `stream` (None: nothing to wait for)."""
        import torch
        b = 0 if self.graph else self.count % self.slots
        k = self.count % self.depth
        self.count += 1
        st = self.streams[k] if self.streams else stream
        if self.streams and stream is not None:
            # deeper pipelines launch on their own streams: whatever produced the frames on the caller's stream (an upload,
            # `decode_depth_device`) has to be finished before k_boxsum reads them
            st.wait_stream(stream)
        if self.pending[b] is not None:
            self.pending[b].wait()                      # (gloo rehearsal only: the staged gather that last used this buffer pair)
            self.pending[b] = None
        if self.graph:
            self.hp.graph_launch(st.cuda_stream)
        else:
            self.hps[k].predict_batch_device(frames_ptr, self.n_local, self.w, self.h, self.intrinsic, self.pose_bufs[b].data_ptr(),
                                             stream=st.cuda_stream)
        if self.collective:
            with torch.cuda.stream(st):
                if self.on_device:
                    # RCCL: the collective is a node of THIS step's stream (a synchronous-mode c10d collective is enqueued under the
                    # current stream and does not block the host), ordered after the step's kernels and before whatever next
                    # touches the buffer pair on that stream -- no work object, no events between streams, no wait before the
                    # next k_boxsum.  With async_op=True every step paid for two cross-stream events and a barrier packet in front
                    # of its first kernel: one MI355X, one rank through RCCL, 713 k frames/s against 737 k this way and 739 k
                    # with no collective at all (tools/experiments/dist_overhead.py, profiles/r03_experiments.md).
                    self.dist.all_gather_into_tensor(self.gathered[b], self.pose_bufs[b], group=self.group)
                else:
                    self.pending[b] = self.dist.all_gather_into_tensor(self.gathered[b], self.pose_bufs[b].cpu(), group=self.group, async_op=True)
        return b

    def fence(self):
        import torch
        for i in range(self.slots):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None
        if self.collective:
            self.dist.barrier(group=self.group)
        torch.cuda.synchronize(self.device)

    def last_poses(self) -> np.ndarray:
        """Gathered pose records (all ranks' shards in rank order) of the most recent step; call after fence()."""
        b = 0 if self.graph else (self.count - 1) % self.slots
        return poses_from_bytes(self.gathered[b] if self.collective else self.pose_bufs[b])
