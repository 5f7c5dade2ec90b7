"""GPU tests added in round 3: the RCCL path executed on one GPU, an imported (serde-JSON) forest through the HIP path,
the parity taps of chunked host batches, BASELINE configs[4] with its stated forest, concurrent predictors, and the
exception containment of the C ABI.

All comparisons are HIP path (through the C ABI) vs the CPU oracle, bit-exact.
"""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from depthhead_amd import biwi, stamm_json, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hp_mod(hip_lib):
    from depthhead_amd import prediction
    return prediction


def _poses_equal(a, ref):
    return np.array_equal(a["mid_point"], ref["mid_point"]) and np.array_equal(a["rotation"], ref["rotation"])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# ------------------------------------------------------------------ the RCCL gather, executed
def test_rccl_all_gather_path_on_one_gpu(oracle, hip_lib, tmp_path):
    """`bench.py --gpus 1 --force-dist` launched by torch.distributed.run with ONE rank: the process group is initialised
    with the `nccl` backend (= RCCL on ROCm) on this box's GPU and every step goes through ShardedPredictor's stream-ordered
    `all_gather_into_tensor` on DEVICE buffers -- the code path the 8-GPU run takes (bench.py's init_process_group,
    dist.py's collective, the MAX all-reduce of the timing, barrier, destroy).  The gathered poses of the last timed step
    equal the oracle's.  A fresh child process, never an exec of this one."""
    dump = str(tmp_path / "poses.npy")
    nf, w, h, trees, depth = 12, 320, 240, 6, 10
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--frames", str(nf),
           "--width", str(w), "--height", str(h), "--trees", str(trees), "--depth", str(depth), "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--no-extras", "--dump-poses", dump]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["value"] > 0
    assert out["config"]["collective"].startswith("RCCL all_gather_into_tensor"), out["config"]
    from depthhead_amd._lib import POSE_DTYPE
    got = np.load(dump)
    assert got.dtype == POSE_DTYPE and got.shape == (nf,)
    forest = synth.fit_forest(trees, depth, synth.FOREST_SEED_BASE + 2)
    model = synth.ModelParams(stepwidth=4)
    K = synth.default_intrinsic(w, h)
    frames = synth.biwi_batch(nf, w, h, first=0) if out["last_step_batch"] == "a" else np.roll(synth.biwi_batch(nf, w, h, first=10000), 7, axis=0)
    assert _poses_equal(got, oracle.predict_batch(forest, model, frames, K))


def test_predict_stream_over_rccl_world_of_one(oracle, hip_lib, tmp_path):
    """dist.predict_stream / gather_poses with the `nccl` backend in a world of one rank (device buffers through RCCL)."""
    script = tmp_path / "stream1.py"
    script.write_text(f"""
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch, torch.distributed as dist
from depthhead_amd import synth
from depthhead_amd.dist import predict_stream
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
forest = synth.fit_forest(4, 8, synth.FOREST_SEED_BASE + 9, n_frames=8, subset=800)
frames = synth.biwi_batch(7, 200, 160, first=70)
with HoughPrediction(forest, synth.ModelParams(stepwidth=4)) as hp:
    poses = predict_stream(hp, frames, IntrinsicMatrix(synth.default_intrinsic(200, 160)))
np.save(sys.argv[1], poses)
dist.barrier(); dist.destroy_process_group()
""")
    dump = str(tmp_path / "p.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script), dump], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    forest = synth.fit_forest(4, 8, synth.FOREST_SEED_BASE + 9, n_frames=8, subset=800)
    ref = oracle.predict_batch(forest, synth.ModelParams(stepwidth=4), synth.biwi_batch(7, 200, 160, first=70), synth.default_intrinsic(200, 160))
    assert _poses_equal(np.load(dump), ref)


# ------------------------------------------------------------------ an imported forest on the HIP path
@pytest.mark.parametrize("one_child", ["right", "left"])
def test_imported_serde_json_forest_through_the_hip_path(hp_mod, oracle, one_child):
    """A fitted forest serialised to the serde-JSON shapes of prediction.rs:239-256 / houghforest.rs:63-78 / types.rs:33-37,
    imported with an explicit `one_child`, run through the HIP path and compared with the oracle ON THE SAME IMPORTED
    FOREST stage by stage (leaf ids, flags, grids, guesses, traces, pose).  The flipped convention is a different model:
    its leaves differ.  PARITY UNPINNED for stamm's nesting and child convention (crate not vendored, Cargo.toml:17)."""
    from test_gpu_parity import _check_frames
    src = synth.fit_forest(5, 9, synth.FOREST_SEED_BASE + 31, n_frames=10, subset=1200)
    params = synth.ModelParams(stepwidth=4, gaussian_sigma=7.5, meanshift_iterations=17)
    text = stamm_json.export_json(src, params, one_child="right")          # the document says: Binar::One -> right
    forest, model = stamm_json.import_json(text, one_child=one_child)
    assert (model.stepwidth, model.gaussian_sigma, model.meanshift_iterations) == (4, 7.5, 17)
    assert forest.n_nodes == src.n_nodes and forest.n_leaves == src.n_leaves
    w, h = 320, 240
    frames = synth.biwi_batch(3, w, h, first=820)
    K = synth.default_intrinsic(w, h)
    poses = _check_frames(hp_mod, oracle, forest, model, frames, K)
    # against the forest the document was written from: same model under "right" (renumbered nodes / leaves), another under "left"
    ref_src = oracle.predict_batch(src, params, frames, K)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.debug_enable(True)
        hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        leaf = hp.debug_leaf_indices(3, w, h)
    with hp_mod.HoughPrediction(src, params, device=0) as hp:
        hp.debug_enable(True)
        hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        leaf_src = hp.debug_leaf_indices(3, w, h)
    walked = leaf_src >= 0
    same_leaf_payload = np.array_equal(forest.leaf_prob[leaf[walked]], src.leaf_prob[leaf_src[walked]])
    if one_child == "right":
        assert _poses_equal(poses, ref_src) and same_leaf_payload
    else:
        assert not same_leaf_payload                                          # the flipped convention walks to other leaves


# ------------------------------------------------------------------ taps of a chunked host batch
def test_taps_of_a_host_batch_larger_than_one_upload_chunk(hp_mod, oracle):
    """With the taps on, a host batch is ONE device batch whatever its size (round 2 split 40 frames into 20 + 20 and the
    taps then described the second half under the first frames' indices): 40 frames, every stage against the oracle."""
    from test_gpu_parity import _check_frames
    forest = synth.fit_forest(4, 8, synth.FOREST_SEED_BASE + 9, n_frames=8, subset=800)
    frames = synth.biwi_batch(40, 160, 120, first=1200)
    _check_frames(hp_mod, oracle, forest, synth.ModelParams(stepwidth=6), frames, synth.default_intrinsic(160, 120), full=False)
    # product mode (taps off): the same call is chunked (16 + 24, ...) and gives the same poses; the grid / hit-count taps
    # then describe the last chunk only (include/depthhead_hip.h)
    os.environ["DH_STAGE_CHUNK"] = "16"
    try:
        with hp_mod.HoughPrediction(forest, synth.ModelParams(stepwidth=6), device=0) as hp:
            a = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(synth.default_intrinsic(160, 120)))
    finally:
        os.environ.pop("DH_STAGE_CHUNK", None)
    assert _poses_equal(a, oracle.predict_batch(forest, synth.ModelParams(stepwidth=6), frames, synth.default_intrinsic(160, 120)))


# ------------------------------------------------------------------ BASELINE configs[4] with its stated forest
@pytest.fixture(scope="module")
def c5_forests():
    return {"synth": synth.synth_forest(10, 15, synth.FOREST_SEED_BASE + 2), "fitted": synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)}


@pytest.mark.parametrize("kind", ["synth", "fitted"])
def test_config5_stated_forest_stagewise(hp_mod, oracle, c5_forests, kind):
    """320x240 frames, stride 1 (38 400 windows), the 10-tree depth-15 forest -- `synth_forest(10, 15)`
    as BASELINE.md section 4 states it, and the fitted one the bench uses.  Four frames stage by stage vs the oracle."""
    from test_gpu_parity import _check_frames
    frames = synth.biwi_batch(4, 320, 240, first=30)
    _check_frames(hp_mod, oracle, c5_forests[kind], synth.ModelParams(stepwidth=1), frames, synth.default_intrinsic(320, 240), full=False)


@pytest.mark.parametrize("kind", ["synth", "fitted"])
def test_config5_stated_forest_graph_replay_64_frames(hp_mod, oracle, c5_forests, kind):
    """... and the per-GPU form of configs[4]: a 64-frame batch captured once into a hipGraph and replayed (second replay on
    other frames written into the same buffer), poses vs oracle.predict_batch."""
    torch = pytest.importorskip("torch")
    w, h, n = 320, 240, 64
    forest, model = c5_forests[kind], synth.ModelParams(stepwidth=1)
    K = synth.default_intrinsic(w, h)
    a_np, b_np = synth.biwi_batch(n, w, h, first=0), synth.biwi_batch(n, w, h, first=500)
    dev = torch.device("cuda", 0)
    buf = torch.from_numpy(a_np.view(np.int16)).to(dev)
    out = torch.zeros(n * 40, dtype=torch.uint8, device=dev)
    from depthhead_amd._lib import POSE_DTYPE
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.reserve(n, w, h)
        hp.graph_capture(buf.data_ptr(), n, w, h, hp_mod.IntrinsicMatrix(K), out.data_ptr())
        st = torch.cuda.current_stream(dev)
        hp.graph_launch(st.cuda_stream)
        torch.cuda.synchronize()
        pa = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE).copy()
        buf.copy_(torch.from_numpy(b_np.view(np.int16)))
        hp.graph_launch(st.cuda_stream)
        torch.cuda.synchronize()
        pb = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE).copy()
    assert _poses_equal(pa, oracle.predict_batch(forest, model, a_np, K))
    assert _poses_equal(pb, oracle.predict_batch(forest, model, b_np, K))


# ------------------------------------------------------------------ concurrent predictors (include/depthhead_hip.h: "may run concurrently")
def test_concurrent_predictors_keep_their_poses(hp_mod, oracle):
    """Host threads, each with its own predictor, mixing the host entry points (pageable frames of changing batch sizes --
    workspaces grow beside other predictors' running kernels --, run-length coded payloads, predict_mask, guesses); every
    result is compared with a single-threaded reference pass.  One deterministic sequence per thread (tools/soak_threads.py
    is the long form of this)."""
    w, h = 320, 240
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 81, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=3)
    frames = synth.biwi_batch(30, w, h, first=900)
    intr = hp_mod.IntrinsicMatrix(synth.default_intrinsic(w, h))
    pay = [biwi.encode_depth(f) for f in frames]
    rs0 = np.random.RandomState(5)
    midp = rs0.uniform(-200, 1200, (30, 3)).astype(np.float32)
    rot = rs0.uniform(-1, 1, (30, 3))
    with hp_mod.HoughPrediction(forest, model) as hp0:
        ref = hp0.predict_batch(frames, intr).copy()
        refg = hp0.predict_batch(frames, intr, midp, rot).copy()
        rmask = hp0.predict_mask(frames[:3]).copy()
    assert _poses_equal(ref[:8], oracle.predict_batch(forest, model, frames[:8], synth.default_intrinsic(w, h)))
    errs = []

    def work(t):
        try:
            rs = np.random.RandomState(100 + t)
            with hp_mod.HoughPrediction(forest, model) as hp:
                for it in range(18):
                    idx = rs.randint(0, 30, int(rs.randint(1, 31)))
                    k = (it + t) % 4
                    if k == 0:
                        out, want = hp.predict_batch(frames[idx].copy(), intr), ref[idx]
                    elif k == 1:
                        out, want = hp.predict_batch_rle([pay[i] for i in idx], intr), ref[idx]
                    elif k == 2:
                        out, want = hp.predict_batch(frames[idx].copy(), intr, midp[idx], rot[idx]), refg[idx]
                    else:
                        if not np.array_equal(hp.predict_mask(frames[:3]), rmask):
                            errs.append((t, it, "predict_mask"))
                        continue
                    if not _poses_equal(out, want):
                        errs.append((t, it, k, len(idx)))
        except Exception as e:   # noqa
            errs.append((t, repr(e)))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs


# ------------------------------------------------------------------ nothing throws across the ABI
def test_bad_payload_batches_come_back_as_error_codes(hp_mod, hip_lib):
    """Every `extern "C"` entry point runs inside dh_guard_ (dh_host.h; the guard itself -- bad_alloc, std::exception, anything
    else -> an error code -- is exercised under the sanitizers by tests/host/host_check.cpp).  Here, through the real library:
    malformed batches are error codes with the frame named, and the predictor stays usable afterwards."""
    import ctypes as C
    from depthhead_amd._lib import DepthheadError
    forest = synth.fit_forest(4, 8, synth.FOREST_SEED_BASE + 9, n_frames=8, subset=800)
    with hp_mod.HoughPrediction(forest, synth.ModelParams(stepwidth=4), device=0) as hp:
        good = biwi.encode_depth(synth.biwi_like(96, 96, 5))
        buf = np.frombuffer(good, dtype=np.uint8)
        n = 3
        ptrs = (C.c_void_p * n)(*([buf.ctypes.data] * n))
        lens = (C.c_size_t * n)(*([buf.size] * n))
        w_, h_ = C.c_uint32(), C.c_uint32()
        assert hip_lib.dh_biwi_decode_depth_device(hp._ph, ptrs, lens, C.c_int(n), None, C.c_size_t(0), C.byref(w_), C.byref(h_)) == 0
        assert (w_.value, h_.value) == (96, 96)
        # a NULL payload inside the batch
        ptrs2 = (C.c_void_p * n)(buf.ctypes.data, None, buf.ctypes.data)
        rc = hip_lib.dh_biwi_decode_depth_device(hp._ph, ptrs2, lens, C.c_int(n), None, C.c_size_t(0), C.byref(w_), C.byref(h_))
        assert rc == -1 and b"frame 1" in hip_lib.dh_last_error()
        with pytest.raises(DepthheadError):
            hp.predict_batch_rle([good, good[:11]], hp_mod.IntrinsicMatrix(synth.default_intrinsic(96, 96)))
        # the predictor is still usable
        out = hp.predict_batch_rle([good], hp_mod.IntrinsicMatrix(synth.default_intrinsic(96, 96)))
        assert out.shape == (1,)


# ------------------------------------------------------------------ profiling: HIP events + roctx ranges
def test_profiling_changes_nothing_and_times_every_kernel(hp_mod, oracle):
    """dh_set_profiling: HIP events around every kernel and roctx ranges around every launch (the marker library is looked up at
    run time; absent, the ranges are no-ops).  Same poses with it on, every kernel's duration positive and inside the total;
    dh_get_timing before any profiled batch is DH_ESTATE."""
    from depthhead_amd._lib import DepthheadError
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 9, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=4)
    frames = synth.biwi_batch(6, 320, 240, first=40)
    K = synth.default_intrinsic(320, 240)
    ref = oracle.predict_batch(forest, model, frames, K)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        with pytest.raises(DepthheadError) as ei:
            hp.timing()
        assert ei.value.code == -6
        a = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        hp.set_profiling(True)
        b = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        t = hp.timing()
        hp.set_profiling(False)
        c = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
    assert _poses_equal(a, ref) and _poses_equal(b, ref) and _poses_equal(c, ref)
    parts = [t[k] for k in ("boxsum_ms", "traverse_ms", "emit_ms", "vote_ms", "cluster_ms")]
    assert all(p > 0.0 for p in parts) and t["n_frames"] == 6
    assert abs(sum(parts) - t["total_ms"]) < 0.02 * t["total_ms"] + 1e-3


# ------------------------------------------------------------------ k_region's blocks: windows that travel
@pytest.mark.parametrize("leaf_hist", [True, False])
def test_windows_that_travel_inside_and_beyond_the_blocks_of_k_region(hp_mod, oracle, leaf_hist):
    """Small batches with many hit records: k_region builds 64^3 cells of the position accumulator and 32^3 of the rotation
    accumulator around the initial guesses, and k_cluster cuts every 26^3 region it needs out of them -- centred on the window
    while the block allows, pushed back inside it near its faces -- and gathers a region from the hit records again only once
    the window has left the block.  A dense forest at stride 1 with 45 iterations (the position window drifts a cell per
    iteration: 31 cells on the first frame) and rotation guesses 5 and 9 bins off take all three ways; every iteration of both
    mean shifts against the oracle's trace, then the plain product path."""
    from test_gpu_parity import _check_frames
    forest = synth.synth_forest(16, 10, synth.FOREST_SEED_BASE + 3)
    model = synth.ModelParams(stepwidth=1, meanshift_iterations=45)
    w, h = 640, 480
    K = synth.default_intrinsic(w, h)
    frames = np.stack([synth.biwi_like(w, h, synth.FRAME_SEED_BASE + s) for s in (16, 12)])
    base = [oracle.predict(forest, model, f, K) for f in frames]
    rot = np.stack([base[0].rotation + np.array([5, 0, -5]) * 3.14159 / 60, base[1].rotation + np.array([-9, 0, 9]) * 3.14159 / 60])
    mask = np.array([2, 2], dtype=np.uint8)                       # rotation guesses only: the position windows start at their own guess
    ref = [oracle.predict(forest, model, frames[i], K, None, rot[i]) for i in range(2)]
    travel = lambda tr: int(np.abs(np.asarray(tr, dtype=np.int64) - np.asarray(tr[0], dtype=np.int64)).max())
    # the workload does what the test is for (block half-widths: 19 + 3 cells for the position, 3 + 3 for the rotation)
    assert travel(ref[0].ms_trace_mid) > 22 and 3 < travel(ref[0].ms_trace_rot) <= 6 and travel(ref[1].ms_trace_rot) > 6
    env = {"DH_REGION_MIN_HITS": "1"}
    if not leaf_hist:
        env["DH_NO_LEAF_HIST"] = "1"
    os.environ.update(env)
    try:
        _check_frames(hp_mod, oracle, forest, model, frames, K, None, rot, mask, full=False)
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            for _ in range(2):                                    # the blocks are zeroed per batch
                got = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K), None, rot, mask)
                for i in range(2):
                    assert np.array_equal(got["mid_point"][i], ref[i].mid_point) and np.array_equal(got["rotation"][i], ref[i].rotation)
    finally:
        for k in env:
            os.environ.pop(k, None)


# ------------------------------------------------------------------ rotation bins with more than 255 votes
@pytest.mark.parametrize("leaf_hist", [True, False])
def test_leaves_with_hundreds_of_equal_rotations(hp_mod, oracle, leaf_hist):
    """k_leaf_prepare keeps a leaf's DISTINCT rotation bins with the number of votes in the bin's top byte: a bin that collects
    more than 255 of the leaf's votes becomes several entries (255 + 255 + ...).  Three single-split trees whose leaves hold
    700 / 256 / 255 / 300 + 300 / 1 / 511 rotations, most of them equal: the full rotation accumulator (every cell and value)
    and both mean shifts against the oracle, through the leaf histogram and through the rotation records."""
    from depthhead_amd.forest import Forest, NODE_DTYPE
    from test_gpu_parity import _check_frames
    nodes = np.zeros(3, dtype=NODE_DTYPE)
    nodes[0] = ((4, 4, 28, 28), (40, 40, 64, 64), 0.0, ~0, ~1)
    nodes[1] = ((10, 30, 34, 54), (44, 6, 68, 30), 10.0, ~2, ~3)
    nodes[2] = ((0, 0, 24, 24), (56, 56, 80, 80), -5.0, ~4, ~5)
    roots = np.array([0, 1, 2], dtype=np.int32)
    prob = np.array([1.0, 0.9, 0.95, 1.0, 0.8, 1.0])
    n = [700, 256, 255, 600, 1, 511]
    begin = np.concatenate([[0], np.cumsum(n)]).astype(np.uint32)
    rs = np.random.RandomState(11)
    centre = rs.uniform(-20, 20, (6, 3))
    rotations = np.repeat(centre, n, axis=0)
    rotations[int(begin[3]) + 300:int(begin[4])] += 3.0              # leaf 3: two bins of 300 votes each
    rotations[int(begin[0]):int(begin[0]) + 40] += rs.uniform(-6, 6, (40, 3))   # leaf 0: 660 equal votes and 40 scattered ones
    offsets = (np.repeat(rs.uniform(-30, 30, (6, 3)), n, axis=0) + rs.uniform(-1, 1, (int(begin[-1]), 3))).astype(np.float32)
    forest = Forest(roots, nodes, prob, begin, begin.copy(), offsets, rotations)
    base = synth.biwi_like(640, 480, 777)
    ys, xs = np.nonzero(base)
    cy, cx = int(ys.mean()), int(xs.mean())
    frames = np.stack([base[cy - 60:cy + 60, cx - 70:cx + 70], base[cy - 40:cy + 80, cx - 90:cx + 50]]).copy()
    model = synth.ModelParams(stepwidth=4)
    K = synth.default_intrinsic(140, 120)
    env = {} if leaf_hist else {"DH_NO_LEAF_HIST": "1"}
    os.environ.update(env)
    try:
        _check_frames(hp_mod, oracle, forest, model, frames, K, full=True)
        for extra in ({"DH_REGION_MIN_HITS": "1"},):                  # ... and through k_region's blocks
            os.environ.update(extra)
            try:
                _check_frames(hp_mod, oracle, forest, model, frames, K, full=True)
            finally:
                for k in extra:
                    os.environ.pop(k, None)
    finally:
        for k in env:
            os.environ.pop(k, None)


# ------------------------------------------------------------------ single-frame workspaces: one-pass tiles
@pytest.mark.parametrize("w,h,stride,trees", [(320, 240, 1, 10), (640, 480, 4, 10), (320, 240, 2, 6), (200, 160, 1, 20)])
def test_single_frame_workspace_takes_one_pass_tiles(hp_mod, oracle, w, h, stride, trees):
    """A workspace reserved for ONE frame (the live-camera loop, examples/live_prediction.rs:76-86) gets tiles of at most
    3200 / trees windows -- every walk of a tile in one lock-step pass -- instead of the largest tile that fits LDS; a workspace for
    more frames keeps the large tiles.  Every stage of the single frame against the oracle, then the same predictor grown to
    three frames."""
    from test_gpu_parity import _check_frames
    forest = synth.fit_forest(trees, 12, synth.FOREST_SEED_BASE + 31, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=stride)
    K = synth.default_intrinsic(w, h)
    frames = synth.biwi_batch(3, w, h, first=60)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.reserve(1, w, h)
        g1 = hp.debug_geometry()
        one = hp.predict_batch(frames[:1], hp_mod.IntrinsicMatrix(K))
        hp.reserve(3, w, h)
        g3 = hp.debug_geometry()
        three = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
    assert g1["px"] * g1["py"] * trees <= 3200 and g3["px"] * g3["py"] >= g1["px"] * g1["py"]
    ref = oracle.predict_batch(forest, model, frames, K)
    assert _poses_equal(one, ref[:1]) and _poses_equal(three, ref)
    _check_frames(hp_mod, oracle, forest, model, frames[:1], K, full=False)


# ------------------------------------------------------------------ tile-flag tags across their wrap-around
def test_tile_flag_tags_wrap_around(hp_mod, oracle):
    """The tile flags carry a per-batch tag (1 .. 255, moved on by k_emit) instead of being zeroed per batch, and k_boxsum clears
    the batch's other counters itself: a stale tag can only be a false positive (a tile walked for nothing).  280 batches through
    ONE predictor -- past the wrap -- alternating frames whose occupied tiles differ (a subject left, right, none at all), every
    pose against the oracle's; the same with the fill dispatch (DH_NO_ZERO_FOLD) and through a replayed hipGraph."""
    torch = pytest.importorskip("torch")
    from depthhead_amd._lib import POSE_DTYPE
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 9, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=4)
    w, h = 320, 240
    K = synth.default_intrinsic(w, h)
    base = synth.biwi_batch(3, w, h, first=40)
    frames = [base[0], np.roll(base[1], 90, axis=1), np.zeros((h, w), dtype=np.uint16), np.roll(base[2], -80, axis=1)]
    ref = [oracle.predict_batch(forest, model, f[None], K)[0] for f in frames]
    intr = hp_mod.IntrinsicMatrix(K)
    for env in ({}, {"DH_NO_ZERO_FOLD": "1"}):
        os.environ.update(env)
        try:
            with hp_mod.HoughPrediction(forest, model, device=0) as hp:
                for i in range(280):
                    k = (i * 7 + i // 5) % 4
                    got = hp.predict_batch(frames[k][None].copy(), intr)
                    assert np.array_equal(got["mid_point"][0], ref[k]["mid_point"]) and np.array_equal(got["rotation"][0], ref[k]["rotation"]), (i, k, env)
        finally:
            for kk in env:
                os.environ.pop(kk, None)
    dev = torch.device("cuda:0")
    fr = torch.zeros((1, h, w), dtype=torch.int16, device=dev)
    out = torch.zeros(POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.reserve(1, w, h)
        hp.graph_capture(fr.data_ptr(), 1, w, h, intr, out.data_ptr())
        for i in range(270):
            k = (i * 3 + i // 7) % 4
            fr.copy_(torch.from_numpy(frames[k][None].view(np.int16)))
            hp.graph_launch(st.cuda_stream)
            st.synchronize()
            got = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
            assert np.array_equal(got["mid_point"][0], ref[k]["mid_point"]) and np.array_equal(got["rotation"][0], ref[k]["rotation"]), ("graph", i, k)


# ------------------------------------------------------------------ forked sub-batches: one tile-flag tag per kernel sequence
@pytest.mark.parametrize("chunks", ["2", "4"])
def test_forked_sub_batches_keep_their_own_tags(hp_mod, oracle, chunks):
    """DH_CHUNKS forks a device batch into sub-batches on streams of their own (automatic from 512 frames): every kernel sequence in
    flight has its own tile-flag tag and the batch keeps its fill dispatch.  96 device-resident frames, several calls through one
    predictor (alternating with unforked calls, whose k_boxsum clears the counters itself), every pose against the oracle's."""
    torch = pytest.importorskip("torch")
    from depthhead_amd._lib import POSE_DTYPE
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 9, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=4)
    w, h, n = 320, 240, 96
    K = synth.default_intrinsic(w, h)
    base = synth.biwi_batch(24, w, h, first=70)
    base[5] = 0
    frames = base[np.arange(n) % 24]
    ref = oracle.predict_batch(forest, model, base, K)[np.arange(n) % 24]
    dev = torch.device("cuda:0")
    fr = torch.from_numpy(frames.view(np.int16)).to(dev)
    out = torch.zeros(n * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    intr = hp_mod.IntrinsicMatrix(K)
    os.environ["DH_CHUNKS"] = chunks
    try:
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            for rep in range(6):
                m = n if rep % 2 == 0 else 24                     # 24 frames: below two chunks' minimum, one sequence
                hp.predict_batch_device(fr.data_ptr(), m, w, h, intr, out.data_ptr(), stream=st.cuda_stream)
                st.synchronize()
                got = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE)[:m]
                assert _poses_equal(got, ref[:m]), (chunks, rep)
    finally:
        os.environ.pop("DH_CHUNKS", None)


@pytest.mark.gpu
def test_set_forking_and_batches_beyond_one_round_of_boxsum_workgroups(hp_mod, oracle):
    """dh_predictor_set_forking (0 automatic / 1 never / n forced) and k_boxsum's band rule (dh_box_bands_) for batches whose
    workgroups do not fit the chip at once: 257 ... 512 device-resident 640 x 480 frames in ONE kernel sequence (320 and 512: 5 and 4
    bands per frame instead of the 3 and 2 the wave-count rule took), forked in two and three, and automatic; every pose against the oracle's
    (the frames are replicas of 16 distinct ones).  Arguments outside 0 .. 8 are refused."""
    torch = pytest.importorskip("torch")
    from depthhead_amd._lib import POSE_DTYPE, DepthheadError
    forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
    model = synth.ModelParams(stepwidth=4)
    w, h = 640, 480
    K = synth.default_intrinsic(w, h)
    base = synth.biwi_batch(16, w, h, first=420)
    base[3] = 0
    ref16 = oracle.predict_batch(forest, model, base, K)
    dev = torch.device("cuda:0")
    intr = hp_mod.IntrinsicMatrix(K)
    st = torch.cuda.current_stream(dev)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        for bad in (-1, 9):
            with pytest.raises(DepthheadError):
                hp.set_forking(bad)
        for n in (257, 320, 449, 512):             # 5 / 5 / 4 / 4 bands per frame
            idx = (np.arange(n) * 7) % 16
            fr = torch.from_numpy(base[idx].view(np.int16)).to(dev)
            out = torch.zeros(n * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            for chunks in (1, 2, 0, 3, 1):
                hp.set_forking(chunks)
                out.zero_()
                hp.predict_batch_device(fr.data_ptr(), n, w, h, intr, out.data_ptr(), stream=st.cuda_stream)
                st.synchronize()
                got = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
                assert _poses_equal(got, ref16[idx]), (n, chunks)
            del fr, out


@pytest.mark.gpu
@pytest.mark.parametrize("bands", ["1", "3", "7", "99"])
def test_forced_boxsum_bands_do_not_change_results(hp_mod, oracle, bands):
    """DH_BOX_BANDS (the experiments' switch for the number of bands k_boxsum cuts a frame into, LDS-ring path): whatever the cut --
    one band, bands that do not divide the rows, more bands than mask blocks (clamped) -- every pose matches the oracle's, also on
    the second batch through the same workspace (the zero-skip masks of the first cut's blocks are reused)."""
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 9, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=4)
    w, h = 320, 240
    K = synth.default_intrinsic(w, h)
    a = synth.biwi_batch(12, w, h, first=130)
    b = synth.biwi_batch(12, w, h, first=500)
    b[4] = 0
    ra, rb = oracle.predict_batch(forest, model, a, K), oracle.predict_batch(forest, model, b, K)
    os.environ["DH_BOX_BANDS"] = bands
    try:
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            intr = hp_mod.IntrinsicMatrix(K)
            for frames, ref in ((a, ra), (b, rb), (a, ra)):
                assert _poses_equal(hp.predict_batch(frames, intr), ref), bands
    finally:
        os.environ.pop("DH_BOX_BANDS", None)
