"""The N>1 path on CPU: two `gloo` ranks shard a frame stream, each computes its shard (the CPU
oracle stands in for the device path, which needs a GPU) and the pose records are all-gathered with
the same `depthhead_amd.dist.gather_poses` the GPU driver uses over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util
from depthhead_amd.dist import POSE_BYTES, gather_poses, poses_from_bytes, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as po
        forest, model, frames, K, _ = golden_util.load("tiny_96x96_s4")
        stream = np.concatenate([frames] * ((n_total + len(frames) - 1) // len(frames)))[:n_total]
        a, b = shard_range(n_total, rank, world)
        local = po.predict_batch(forest, model, stream[a:b], K, threads=1) if b > a else np.zeros(0, dtype=po.POSE_DTYPE)
        buf = torch.from_numpy(np.frombuffer(local.tobytes(), dtype=np.uint8).copy())
        allp = gather_poses(buf, n_total)
        assert allp.numel() == n_total * POSE_BYTES
        np.save(os.path.join(outdir, f"rank{rank}.npy"), allp.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [6, 5])          # even and ragged shards
def test_two_rank_shard_and_gather(tmp_path, oracle, n_total):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    forest, model, frames, K, exp = golden_util.load("tiny_96x96_s4")
    got = [poses_from_bytes(np.load(tmp_path / f"rank{r}.npy")) for r in range(world)]
    assert got[0].tobytes() == got[1].tobytes()       # every rank ends with the same gathered stream
    for i in range(n_total):
        e = exp[i % len(exp)]
        assert np.array_equal(got[0]["mid_point"][i], e["mid_point"]) and np.array_equal(got[0]["rotation"][i], e["rotation"])


class _OracleStandIn:
    """Stands where a rank's GPU predictor stands in `predict_stream` (this suite has no GPU): same
    `predict_batch(frames, intrinsic)` signature, computed by the CPU oracle."""

    def __init__(self, forest, model):
        self.forest, self.model = forest, model

    def predict_batch(self, frames, intrinsic):
        from oracle import pyoracle as po
        return po.predict_batch(self.forest, self.model, frames, intrinsic, threads=1)


def _stream_worker(rank, world, port, n_total, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from depthhead_amd.dist import predict_stream
        forest, model, frames, K, _ = golden_util.load("tiny_96x96_s4")
        stream = np.concatenate([frames] * ((n_total + len(frames) - 1) // len(frames)))[:n_total]
        poses = predict_stream(_OracleStandIn(forest, model), stream, K)
        assert poses.shape == (n_total,)
        np.save(os.path.join(outdir, f"stream{rank}.npy"), np.frombuffer(poses.tobytes(), dtype=np.uint8))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 1])          # ragged shards; one rank with an empty shard
def test_predict_stream_two_ranks(tmp_path, oracle, n_total):
    """`depthhead_amd.dist.predict_stream` -- the entry point a multi-GPU user drives -- on two gloo ranks."""
    world, port = 2, _free_port()
    mp.spawn(_stream_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    forest, model, frames, K, exp = golden_util.load("tiny_96x96_s4")
    got = [poses_from_bytes(np.load(tmp_path / f"stream{r}.npy")) for r in range(world)]
    assert got[0].tobytes() == got[1].tobytes()
    for i in range(n_total):
        e = exp[i % len(exp)]
        assert np.array_equal(got[0]["mid_point"][i], e["mid_point"]) and np.array_equal(got[0]["rotation"][i], e["rotation"])


def test_predict_stream_without_a_process_group(oracle):
    from depthhead_amd.dist import predict_stream
    forest, model, frames, K, exp = golden_util.load("tiny_96x96_s4")
    poses = predict_stream(_OracleStandIn(forest, model), frames, K)
    for i, e in enumerate(exp):
        assert np.array_equal(poses["mid_point"][i], e["mid_point"]) and np.array_equal(poses["rotation"][i], e["rotation"])
