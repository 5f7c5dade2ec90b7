"""Row N3 on the device: BIWI run-length coded depth payloads (src/db_reader/biwi.rs:81-103) uploaded as they are
and decoded by k_rle_decode, byte-exact against the host decoder of the C ABI (dh_biwi_decode_depth) and the
pure-Python oracle restatement; dh_predict_batch_rle against dh_predict_batch and the oracle; malformed payloads are
refused with DH_EINVAL before anything is launched.  PARITY UNPINNED for the format itself: the reference holds no
sample `.bin` (the database cannot be downloaded here); payloads come from `biwi.encode_depth`."""
import os
import struct

import numpy as np
import pytest

from depthhead_amd import biwi, synth
from oracle import biwi_oracle as bo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp_mod(hip_lib):
    from depthhead_amd import prediction
    return prediction


def _small_predictor(hp_mod, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        forest = synth.fit_forest(4, 8, synth.FOREST_SEED_BASE + 9, n_frames=8, subset=800)
        return hp_mod.HoughPrediction(forest, synth.ModelParams(stepwidth=4), device=0), forest
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _frames(n, w, h, seed):
    rs = np.random.RandomState(seed)
    out = np.stack([synth.biwi_like(max(w, 96), max(h, 96), synth.FRAME_SEED_BASE + seed * 100 + i)[:h, :w] for i in range(n)]).copy()
    if n > 2:
        out[1] = 0                                                       # one run of w*h empty pixels
        out[2] = rs.randint(1, 65536, (h, w))                            # one run of w*h values
    if n > 3:
        out[3] = (rs.rand(h, w) < 0.5) * rs.randint(1, 65536, (h, w))    # a run every other pixel
    if n > 4:
        out[4, :, -1] = 7; out[4, -1, :] = 9                             # runs that end exactly at row / image ends
    return out.astype(np.uint16)


@pytest.mark.parametrize("n,w,h", [(70, 320, 240), (5, 640, 480), (6, 17, 5), (3, 1, 1), (1, 96, 96)])
def test_device_decode_is_byte_exact(hp_mod, n, w, h):
    torch = pytest.importorskip("torch")
    frames = _frames(n, w, h, 3 + n)
    payloads = [biwi.encode_depth(f) for f in frames]
    hp, _ = _small_predictor(hp_mod)
    with hp:
        assert hp.decode_depth_device(payloads) == (w, h)                 # size query + validation only
        dev = torch.full((n, h, w), 0x5a5a, dtype=torch.int16, device="cuda:0")
        assert hp.decode_depth_device(payloads, dev.data_ptr(), dev.numel()) == (w, h)
        got = dev.cpu().numpy().view(np.uint16)
    assert np.array_equal(got, frames)
    for i in (0, n // 2, n - 1):
        assert np.array_equal(got[i], biwi.read_depth(payloads[i]))       # the host decoder of the C ABI
        if w * h <= 320 * 240:
            assert np.array_equal(got[i], bo.read_depth(payloads[i]))     # the oracle restatement of biwi.rs:81-103


def test_payloads_with_redundant_runs(hp_mod):
    """Encodings read_depth accepts but an encoder would not write: zero-length runs, a non-empty run split in two,
    trailing bytes after the last run."""
    torch = pytest.importorskip("torch")
    w, h = 8, 3
    px = np.arange(1, 25, dtype=np.uint16)
    def run(e, vals):
        return struct.pack("<II", e, len(vals)) + np.asarray(vals, dtype="<u2").tobytes()
    blob = struct.pack("<II", w, h) + run(0, []) + run(2, px[:5]) + run(0, px[5:7]) + run(0, []) + run(3, px[7:19]) + run(2, []) + b"\xff" * 7
    expect = bo.read_depth(blob)
    assert np.array_equal(biwi.read_depth(blob), expect)
    hp, _ = _small_predictor(hp_mod)
    with hp:
        dev = torch.zeros((2, h, w), dtype=torch.int16, device="cuda:0")
        hp.decode_depth_device([blob, biwi.encode_depth(expect)], dev.data_ptr(), dev.numel())
        got = dev.cpu().numpy().view(np.uint16)
    assert np.array_equal(got[0], expect) and np.array_equal(got[1], expect)


def test_malformed_payloads_are_refused_before_launch(hp_mod):
    torch = pytest.importorskip("torch")
    from depthhead_amd._lib import DepthheadError
    w, h = 96, 96
    good = biwi.encode_depth(_frames(1, w, h, 11)[0])
    hdr = struct.pack("<II", w, h)
    bad = {
        "truncated header": good[:6],
        "truncated run header": good[:8 + 4],
        "truncated run data": good[:len(good) // 2 | 1],
        "empty run overruns": hdr + struct.pack("<II", w * h + 1, 0),
        "full run overruns": hdr + struct.pack("<II", w * h - 1, 2) + b"\x01\x00\x02\x00",
        "other size": biwi.encode_depth(_frames(1, 100, 96, 12)[0]),
        "no runs at all": hdr,
    }
    hp, _ = _small_predictor(hp_mod)
    with hp:
        dev = torch.full((3, h, w), 0x1234, dtype=torch.int16, device="cuda:0")
        for name, blob in bad.items():
            for batch in ([good, blob, good], [good, good, blob]):
                with pytest.raises(DepthheadError) as ei:
                    hp.decode_depth_device(batch, dev.data_ptr(), dev.numel())
                assert ei.value.code in (-1, -5), name
                assert f"frame {batch.index(blob)}" in str(ei.value), (name, str(ei.value))
                with pytest.raises(DepthheadError):
                    hp.predict_batch_rle(batch, hp_mod.IntrinsicMatrix(synth.default_intrinsic(w, h)))
        torch.cuda.synchronize()
        assert bool((dev == 0x1234).all())                               # nothing was launched: the output is untouched
        # the host decoder refuses the same payloads (same checks)
        for name, blob in bad.items():
            if name != "other size":
                with pytest.raises(DepthheadError):
                    biwi.read_depth(blob)


@pytest.mark.parametrize("slice_frames", [None, 4])
def test_predict_batch_rle_matches_raw_frames_and_oracle(hp_mod, oracle, slice_frames):
    w, h, n = 320, 240, 11
    frames = synth.biwi_batch(n, w, h, first=400)
    payloads = [biwi.encode_depth(f) for f in frames]
    K = synth.default_intrinsic(w, h)
    intr = hp_mod.IntrinsicMatrix(K)
    rs = np.random.RandomState(2)
    midp = rs.uniform(-200, 1200, (n, 3)).astype(np.float32)
    rot = rs.uniform(-1, 1, (n, 3))
    mask = rs.randint(0, 4, n).astype(np.uint8)
    env = {"DH_STAGE_CHUNK": 2}
    if slice_frames:
        env["DH_MAX_RESIDENT_FRAMES"] = slice_frames
    hp, forest = _small_predictor(hp_mod, **env)
    with hp:
        a = hp.predict_batch_rle(payloads, intr)
        b = hp.predict_batch(frames, intr)
        c = hp.predict_batch_rle(payloads, intr, midp, rot, mask)
        d = hp.predict_batch(frames, intr, midp, rot, mask)
    assert a.tobytes() == b.tobytes() and c.tobytes() == d.tobytes()
    ref = oracle.predict_batch(forest, synth.ModelParams(stepwidth=4), frames, K)
    assert np.array_equal(a["mid_point"], ref["mid_point"]) and np.array_equal(a["rotation"], ref["rotation"])
    for i in range(n):
        mg = midp[i] if mask[i] & 1 else None
        rg = rot[i] if mask[i] & 2 else None
        r = oracle.predict(forest, synth.ModelParams(stepwidth=4), frames[i], K, mg, rg, taps=False)
        assert np.array_equal(c["mid_point"][i], r.mid_point) and np.array_equal(c["rotation"][i], r.rotation), i


def test_pipelined_host_batches_of_every_chunking(hp_mod, oracle):
    """dh_predict_batch uploads in chunks that overlap the kernels: chunk sizes that divide the batch, do not, exceed it."""
    w, h, n = 200, 160, 13
    frames = synth.biwi_batch(n, w, h, first=500)
    K = synth.default_intrinsic(w, h)
    forest = synth.synth_forest(5, 8, synth.FOREST_SEED_BASE + 180)
    model = synth.ModelParams(stepwidth=6)
    ref = oracle.predict_batch(forest, model, frames, K)
    for chunk in (1, 4, 13, 32):
        os.environ["DH_STAGE_CHUNK"] = str(chunk)
        try:
            with hp_mod.HoughPrediction(forest, model, device=0) as hp:
                for _ in range(2):
                    got = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
                    assert np.array_equal(got["mid_point"], ref["mid_point"]) and np.array_equal(got["rotation"], ref["rotation"]), chunk
        finally:
            os.environ.pop("DH_STAGE_CHUNK", None)


def test_page_locked_host_frames(hp_mod, oracle):
    """dh_host_alloc: frames in page-locked host memory take the asynchronous chunked upload; same poses."""
    from depthhead_amd._lib import pinned_empty
    w, h, n = 320, 240, 9
    frames = synth.biwi_batch(n, w, h, first=600)
    pinned = pinned_empty(frames.shape, np.uint16)
    pinned[...] = frames
    assert pinned.flags["C_CONTIGUOUS"] and pinned.dtype == np.uint16
    K = synth.default_intrinsic(w, h)
    hp, forest = _small_predictor(hp_mod, DH_STAGE_CHUNK=2)
    with hp:
        a = hp.predict_batch(pinned, hp_mod.IntrinsicMatrix(K))
        b = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        pinned[0] = frames[5]                                            # the buffer is the caller's: refill and go again
        c = hp.predict_batch(pinned, hp_mod.IntrinsicMatrix(K))
    ref = oracle.predict_batch(forest, synth.ModelParams(stepwidth=4), frames, K)
    assert a.tobytes() == b.tobytes()
    assert np.array_equal(a["mid_point"], ref["mid_point"]) and np.array_equal(a["rotation"], ref["rotation"])
    assert np.array_equal(c["mid_point"][0], ref["mid_point"][5]) and np.array_equal(c["mid_point"][1:], ref["mid_point"][1:])
    del pinned
