"""GPU tests added in round 2: regressions for the bugs the out-of-suite fuzz found, BASELINE configs at
their stated sizes, and the hardening of the runtime (stale hipGraph, tap gating, profiling knobs).

All comparisons are HIP path (through the C ABI) vs the CPU oracle, bit-exact.
"""
import os

import numpy as np
import pytest

from depthhead_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp_mod(hip_lib):
    from depthhead_amd import prediction
    return prediction


def _poses_equal(a, ref):
    return np.array_equal(a["mid_point"], ref["mid_point"]) and np.array_equal(a["rotation"], ref["rotation"])


# ------------------------------------------------------------------ fuzz case 114070 (edge-tile region copy)
def _fuzz_case(case):
    """tools/fuzz_parity.py's default-mode recipe for one case number (same seeds, same draws)."""
    rs = np.random.RandomState(777000 + case)
    sw, sh = int(rs.randint(16, 97)), int(rs.randint(16, 97))
    w, h = sw + int(rs.randint(1, 420)), sh + int(rs.randint(1, 300))
    step = int(rs.choice([1, 2, 3, 4, 4, 4, 5, 6, 8, 10, 12]))
    trees, depth = int(rs.randint(1, 18)), int(rs.randint(1, 12))
    mixed = rs.rand() < 0.2
    forest = synth.synth_forest(trees, depth, 9000 + case, patch=(sw, sh), rect_scale=float(rs.uniform(0.1, 0.6)),
                                rect_scale_max=float(rs.uniform(0.6, 0.9)) if mixed else None,
                                full_depth=int(rs.randint(0, depth + 1)), p_split=float(rs.uniform(0.4, 0.95)))
    model = synth.ModelParams(stepwidth=step, subimage_width=sw, subimage_height=sh,
                              gaussian_sigma=float(rs.uniform(0.5, 30.0)), meanshift_iterations=int(rs.randint(0, 25)))
    n = int(rs.randint(1, 4))
    frames = np.stack([synth.biwi_like(max(w, 96), max(h, 96), 19000 + case * 7 + i)[:h, :w] for i in range(n)]).copy()
    if rs.rand() < 0.25:
        frames[0] = (rs.rand(h, w) < 0.01) * rs.randint(1, 65536, (h, w))
    if rs.rand() < 0.25:
        frames[-1, :, : w // 2] = 0
    f = float(rs.uniform(200, 900))
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], dtype=np.float32)
    return forest, model, frames.astype(np.uint16), K, (w, h, sw, sh, step, mixed)


def _edge_tile_is_cut(geo, model, w):
    """True when the tiling has a last tile column that is cut by the right frame edge such that it owns fewer
    16-byte groups per plane than a full tile: the case the direct-to-LDS region copy got wrong before fc33f8a
    (it copied all q/4 groups of every plane and so read past the 4 words of slack behind the image's last
    column, k_traverse.hip `q4_tile`)."""
    step, sw = model.stepwidth, model.subimage_width
    lw = sw // 2
    nx = (w - (sw - lw) - lw + step - 1) // step
    cx_last = nx - (geo["tiles_x"] - 1) * geo["px"]
    m = 1 << geo["swz_log2"]
    bw_last = (cx_last - 1) * step + sw - geo["rw"] + 1
    q4_tile = (((bw_last + m - 1) >> geo["swz_log2"]) + 3) >> 2
    return geo["uniform"] == 1 and geo["px"] % 4 == 0 and geo["tiles_x"] > 1 and q4_tile < geo["swz_q"] // 4


def test_fuzz_case_114070_edge_tile_region_copy(hp_mod, oracle):
    """gpurun_out/fuzz2.log (round 1): `MISMATCH case 114070 ... 215x193 patch 38x19 step 1 ... mixed False` -- two
    frames wrong without a fault.  The same seeds rebuild the case; the tiling is asserted to contain the cut edge
    tile, so the test fails by construction on the pre-fix predicate (`lane < pieces` only)."""
    forest, model, frames, K, (w, h, sw, sh, step, mixed) = _fuzz_case(114070)
    assert (w, h, sw, sh, step, mixed) == (215, 193, 38, 19, 1, False)
    n = frames.shape[0]
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.debug_enable(True)
        poses = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        geo = hp.debug_geometry()
        leaf = hp.debug_leaf_indices(n, w, h)
        pg, rg = hp.debug_grids(n)
    assert _edge_tile_is_cut(geo, model, w), geo
    for i in range(n):
        ref = oracle.predict(forest, model, frames[i], K)
        assert np.array_equal(leaf[i], ref.leaf_idx), f"frame {i}: leaf indices"
        assert np.array_equal(pg[i], ref.pos_grid) and np.array_equal(rg[i], ref.rot_grid)
        assert np.array_equal(poses["mid_point"][i], ref.mid_point) and np.array_equal(poses["rotation"][i], ref.rotation)


@pytest.mark.parametrize("w,step,rect_scale", [(215, 1, 0.3), (333, 4, 0.3), (250, 2, 0.25), (258, 8, 0.4)])
def test_cut_edge_tiles_constructed(hp_mod, oracle, w, step, rect_scale):
    """Widths chosen so that the last tile column is cut (asserted from the runtime's own tiling), for every
    de-interleave factor m = 1, 2, 4, 8; the frames carry non-zero pixels up to the right edge and in the last rows,
    so the cut tiles are walked."""
    h = 170
    forest = synth.synth_forest(5, 8, synth.FOREST_SEED_BASE + 500 + step, patch=(40, 32), rect_scale=rect_scale)
    model = synth.ModelParams(stepwidth=step, subimage_width=40, subimage_height=32)
    frames = np.stack([synth.biwi_like(640, 480, 6100 + i)[150:150 + h, 200:200 + w] for i in range(3)]).copy()
    frames[2, :, w - 60:] = np.maximum(frames[2, :, w - 60:], 900)      # right edge fully populated
    frames[1, h - 40:, :] = np.maximum(frames[1, h - 40:, :], 700)      # bottom rows too
    K = synth.default_intrinsic(w, h)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.debug_enable(True)
        poses = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        geo = hp.debug_geometry()
        leaf = hp.debug_leaf_indices(3, w, h)
        flags = hp.debug_patch_flags(3, w, h)
    assert _edge_tile_is_cut(geo, model, w), (geo, w, step)
    for i in range(3):
        ref = oracle.predict(forest, model, frames[i], K)
        assert np.array_equal(leaf[i], ref.leaf_idx), f"frame {i}: leaf indices"
        assert np.array_equal(flags[i], ref.patch_flags)
        assert np.array_equal(poses["mid_point"][i], ref.mid_point) and np.array_equal(poses["rotation"][i], ref.rotation)


# ------------------------------------------------------------------ BASELINE configs at their stated sizes
@pytest.fixture(scope="module")
def forest_c3():
    return synth.synth_forest(50, 20, synth.FOREST_SEED_BASE + 3)     # 1.4 M nodes, 408 MB: nodes spill L2, no leaf histogram


def test_config3_as_stated(hp_mod, oracle, forest_c3):
    """BASELINE configs[2] exactly: 50 trees, depth 20, stride 2, 640x480 -- 56 000 windows x 50 trees = 2.8 M walks per
    frame.  Two frames stage by stage against the oracle (leaf ids, flags, grids, guesses, trace, pose)."""
    from test_gpu_parity import _check_frames
    assert forest_c3.n_trees == 50 and forest_c3.n_leaves > 16384
    model = synth.ModelParams(stepwidth=2)
    frames = synth.biwi_batch(2, 640, 480, first=40)
    _check_frames(hp_mod, oracle, forest_c3, model, frames, synth.default_intrinsic(), full=False)


def test_config3_batch32_properties(hp_mod, oracle, forest_c3):
    """configs[2] at its batch size (32 frames): deterministic, independent of batch order and
    composition, and equal to the oracle on every frame (SAT mode: O(1) per node)."""
    model = synth.ModelParams(stepwidth=2)
    base = synth.biwi_batch(16, 640, 480, first=200)
    frames = np.concatenate([base, base[::-1]])
    perm = np.random.RandomState(3).permutation(32)
    K = synth.default_intrinsic()
    with hp_mod.HoughPrediction(forest_c3, model, device=0) as hp:
        a = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        b = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        c = hp.predict_batch(frames[perm], hp_mod.IntrinsicMatrix(K))
        d = hp.predict_batch(frames[5:8], hp_mod.IntrinsicMatrix(K))
    assert a.tobytes() == b.tobytes()
    assert _poses_equal(c, a[perm]) and _poses_equal(d, a[5:8])
    assert _poses_equal(a[16:], a[:16][::-1])
    ref = oracle.predict_batch(forest_c3, model, base, K)
    assert _poses_equal(a[:16], ref)


@pytest.mark.parametrize("n", [512, 513])
def test_config4_per_rank_load(hp_mod, oracle, n):
    """BASELINE configs[3]'s per-rank load: 512 config-2 frames in ONE call (= the resident-slice limit) and 513
    (slice boundary: 512 + 1), through the host entry point; properties + a 16-frame oracle sample."""
    forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
    model = synth.ModelParams(stepwidth=4)
    base = synth.biwi_batch(32, 640, 480, first=300)
    idx = np.arange(n) % 32
    frames = base[idx]
    K = synth.default_intrinsic()
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        a = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        b = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
    assert a.tobytes() == b.tobytes()
    ref = oracle.predict_batch(forest, model, base[:16], K)
    assert _poses_equal(a[:16], ref)
    assert _poses_equal(a, a[:32][idx])            # every replica of a frame gives the same pose, also across the slice boundary
    assert np.all(a["reserved"] == 0)


# ------------------------------------------------------------------ hardening
def test_stale_graph_is_refused(hp_mod, oracle):
    """A captured batch has workspace pointers baked in; a later call that reallocates the workspace (larger batch)
    must make dh_graph_launch fail with DH_ESTATE instead of replaying onto freed memory.  After a new capture the
    replay matches the oracle again."""
    torch = pytest.importorskip("torch")
    from depthhead_amd._lib import POSE_DTYPE, DepthheadError
    forest = synth.synth_forest(6, 10, synth.FOREST_SEED_BASE + 8)
    model = synth.ModelParams(stepwidth=2)
    w, h, n = 320, 240, 2
    K = synth.default_intrinsic(w, h)
    intr = hp_mod.IntrinsicMatrix(K)
    dev = torch.device("cuda:0")
    a = synth.biwi_batch(n, w, h, first=90)
    big = synth.biwi_batch(4, w, h, first=20)
    fr = torch.from_numpy(a.view(np.int16)).to(dev)
    out = torch.zeros(n * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.graph_capture(fr.data_ptr(), n, w, h, intr, out.data_ptr())
        hp.graph_launch(st.cuda_stream)
        st.synchronize()
        p4 = hp.predict_batch(big, intr)                       # n = 4 > 2: the workspace is reallocated
        assert _poses_equal(p4, oracle.predict_batch(forest, model, big, K))
        with pytest.raises(DepthheadError) as ei:
            hp.graph_launch(st.cuda_stream)
        assert ei.value.code == -6 and "stale" in str(ei.value)
        hp.graph_capture(fr.data_ptr(), n, w, h, intr, out.data_ptr())
        hp.graph_launch(st.cuda_stream)
        st.synchronize()
        poses = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
        assert _poses_equal(poses, oracle.predict_batch(forest, model, a, K))
        # other ways to lose the workspace: another frame size (predict_mask), the debug taps
        hp.predict_mask(synth.biwi_batch(1, 200, 160, first=3))
        with pytest.raises(DepthheadError):
            hp.graph_launch(st.cuda_stream)


def test_rotation_vote_tap_needs_debug(hp_mod):
    """dh_debug_votes(which = 1) reads rotation records that k_emit only writes without the leaf histogram or with
    the taps on: without dh_debug_enable it must return DH_ESTATE, not read uninitialised records."""
    from depthhead_amd._lib import DepthheadError
    forest = synth.fit_forest(4, 8, synth.FOREST_SEED_BASE + 9, n_frames=8, subset=800)
    model = synth.ModelParams(stepwidth=4)
    frames = synth.biwi_batch(2, 320, 240)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.predict_batch(frames, hp_mod.IntrinsicMatrix(synth.default_intrinsic(320, 240)))
        with pytest.raises(DepthheadError) as ei:
            hp.debug_votes(0, 1, cap=1024)
        assert ei.value.code == -6
        hp.debug_votes(0, 0, cap=1 << 16)                       # position votes need no taps
        hp.debug_enable(True)
        hp.predict_batch(frames, hp_mod.IntrinsicMatrix(synth.default_intrinsic(320, 240)))
        assert hp.debug_votes(0, 1).shape[1] == 4


def test_stray_profiling_environment_cannot_change_poses(hp_mod, oracle):
    """The kernel-truncating profiling switches are compiled out of the product library: an inherited
    DH_TRAV_STOP / DH_EMIT_STOP / DH_VOTE_STOP / DH_CL_STOP cannot alter results."""
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 9, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=4)
    w, h = 320, 240
    frames = synth.biwi_batch(3, w, h)
    K = synth.default_intrinsic(w, h)
    ref = oracle.predict_batch(forest, model, frames, K)
    stray = {"DH_TRAV_STOP": "1", "DH_EMIT_STOP": "1", "DH_VOTE_STOP": "1", "DH_CL_STOP": "1", "DH_TRAV_STAMPS": "1"}
    old = {k: os.environ.get(k) for k in stray}
    os.environ.update(stray)
    try:
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            got = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert _poses_equal(got, ref)
    assert np.any(ref["mid_point"] != 0)


# ------------------------------------------------------------------ product mode (no parity taps)
def _product_mode_check(hp_mod, oracle, forest, model, frames, K, env=None):
    """Product mode (dh_debug_enable never called): every tap that works without it -- both guess grids, the
    per-frame hit count, every position vote -- and the pose, against the oracle."""
    n = frames.shape[0]
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            poses = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
            pg, rg = hp.debug_grids(n)
            hits = hp.debug_hit_counts(n)
            votes = [hp.debug_votes(i, 0) for i in range(n)]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    total_hits = 0
    for i in range(n):
        ref = oracle.predict(forest, model, frames[i], K)
        gated = (ref.patch_flags & 2) != 0
        lf = ref.leaf_idx[gated]
        prob = forest.leaf_prob[lf]
        nv = forest.off_begin[lf + 1] - forest.off_begin[lf]
        # a hit record = a (gated window, leaf with prob > 0) pair whose leaf passes a covariance gate; every such leaf has votes
        assert hits[i] <= int(((prob > 0) & (nv > 0)).sum()), f"frame {i}: more hit records than voting leaves of gated windows"
        total_hits += int(hits[i])
        assert np.array_equal(pg[i], ref.pos_grid), f"frame {i}: 20x20 guess grid"
        assert np.array_equal(rg[i], ref.rot_grid), f"frame {i}: 20^3 guess grid"
        assert np.array_equal(hp_mod.aggregate_votes(votes[i]), ref.mid_cells), f"frame {i}: position accumulator"
        assert np.array_equal(poses["mid_point"][i], ref.mid_point) and np.array_equal(poses["rotation"][i], ref.rotation), f"frame {i}: pose"
    return total_hits


@pytest.mark.parametrize("general", [False, True])
def test_product_mode_matches_the_oracle(hp_mod, oracle, general):
    """Most stage-wise tests run with the parity taps on; this one runs the configuration users run (taps off: no dense
    leaf array, rotation votes through the leaf histogram) on both traversal paths."""
    forest = synth.fit_forest(10, 12, synth.FOREST_SEED_BASE + 21, n_frames=16, subset=2500)
    model = synth.ModelParams(stepwidth=4)
    w, h = 400, 300
    frames = synth.biwi_batch(3, w, h, first=50)
    hits = _product_mode_check(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), {"DH_FORCE_GENERAL": "1"} if general else {})
    assert hits > 0                                             # some windows pass the gate: the test sees votes


def test_window_means_at_the_gate(hp_mod, oracle):
    """Leaf probabilities engineered so that window means sit exactly on, one ulp above and one ulp below 0.7
    (prediction.rs:582-584 sums in tree order in f64 and compares with > 0.7)."""
    rs = np.random.RandomState(5)
    forest = synth.synth_forest(10, 6, synth.FOREST_SEED_BASE + 22)
    prob = forest.leaf_prob
    voting = prob > 0
    choices = np.array([0.7, np.nextafter(0.7, 1.0), np.nextafter(0.7, 0.0), 0.75, 0.65, 1.0, 0.7000001, 0.6999999])
    prob[voting] = choices[rs.randint(0, choices.size, int(voting.sum()))]
    model = synth.ModelParams(stepwidth=3)
    w, h = 320, 240
    frames = synth.biwi_batch(3, w, h, first=60)
    _product_mode_check(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h))


def test_leaf_probabilities_outside_the_unit_interval(hp_mod, oracle):
    """A hand-edited model with leaf 'probabilities' above 1 (the reference does not validate them)."""
    forest = synth.synth_forest(6, 6, synth.FOREST_SEED_BASE + 23)
    prob = forest.leaf_prob
    prob[prob > 0] *= 1.9
    assert prob.max() > 1.0
    model = synth.ModelParams(stepwidth=4)
    frames = synth.biwi_batch(2, 320, 240, first=64)
    _product_mode_check(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(320, 240))


# ------------------------------------------------------------------ the N > 1 control flow on the HIP library
def test_two_rank_bench_rehearsal_on_one_gpu(oracle, hip_lib, tmp_path):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), with
    `--backend gloo` so that both ranks share this box's one GPU: the shard -> predict (HIP library) -> all-gather loop
    of depthhead_amd.dist.ShardedPredictor runs end to end, and the gathered poses of both ranks equal the oracle's for
    their shards.  A fresh child process, never an exec of this one."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "poses.npy")
    nf, w, h, trees, depth = 8, 320, 240, 6, 10
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", str(nf),
           "--width", str(w), "--height", str(h), "--trees", str(trees), "--depth", str(depth), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--dump-poses", dump]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    from depthhead_amd._lib import POSE_DTYPE
    got = np.load(dump)
    assert got.dtype == POSE_DTYPE and got.shape == (2 * nf,)
    forest = synth.fit_forest(trees, depth, synth.FOREST_SEED_BASE + 2)
    model = synth.ModelParams(stepwidth=4)
    K = synth.default_intrinsic(w, h)
    for rank in range(2):
        if out["last_step_batch"] == "a":
            frames = synth.biwi_batch(nf, w, h, first=rank * nf)      # bench.py's first batch of that rank
        else:
            frames = np.roll(synth.biwi_batch(nf, w, h, first=10000 + rank * nf), 7, axis=0)   # ... its second, alternating batch
        ref = oracle.predict_batch(forest, model, frames, K)
        assert _poses_equal(got[rank * nf:(rank + 1) * nf], ref), f"rank {rank}"


def test_pinhole_projection_falls_back_for_huge_and_non_finite_votes(hp_mod, oracle):
    """k_vote projects with the pinhole form of the intrinsic matrix (five of nine products are x * 0 / z * 1) only for
    waves whose operands are finite and small; leaves with huge or infinite offset votes (0 * inf = NaN has to propagate
    through space_to_img_coord, types.rs:424-428) must take the general expression.  Same grids, votes and poses."""
    rs = np.random.RandomState(11)
    forest = synth.fit_forest(6, 8, synth.FOREST_SEED_BASE + 41, n_frames=10, subset=1200)
    voting = np.flatnonzero(forest.leaf_prob > 0)
    for L in voting[rs.rand(voting.size) < 0.3]:
        ob, oe = int(forest.off_begin[L]), int(forest.off_begin[L + 1])
        kind = rs.randint(0, 4)
        k = ob + rs.randint(0, oe - ob)
        if kind == 0:
            forest.offsets[k] = [3.0e35, -2.0e33, 1.0e31]          # finite, beyond the fast path's bound
        elif kind == 1:
            forest.offsets[k, 2] = -np.inf                         # nz = +inf
        elif kind == 2:
            forest.offsets[k, 0] = np.inf                          # nx = -inf: 0 * inf = NaN in the general product
        else:
            forest.offsets[k] = [np.nan, 0.0, -50.0]
    model = synth.ModelParams(stepwidth=4)
    w, h = 320, 240
    frames = synth.biwi_batch(3, w, h, first=70)
    _product_mode_check(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h))
    # and with the taps on (rotation records instead of the leaf histogram)
    from test_gpu_parity import _check_frames
    _check_frames(hp_mod, oracle, forest, model, frames[:2], synth.default_intrinsic(w, h), full=True)


@pytest.mark.parametrize("int_test", [True, False])
def test_mixed_rectangles_with_thresholds_on_reachable_differences(hp_mod, oracle, int_test):
    """General path (rectangles of different sizes, c1 != c2): the split test is decided on the integer
    D = s1 C2 - s2 C1 against per-node bounds, inside the band by the reference's f64 divisions.  Thresholds are put exactly
    on k / (C1 C2) -- reachable values of avg1 - avg2 -- and one ulp beside them, plus the saturating cases; also rectangles
    of zero area (average_value_in_rect returns 0.0 there, types.rs:335-338).  With DH_NO_GENERAL_INT every visit divides."""
    forest = synth.synth_forest(7, 9, synth.FOREST_SEED_BASE + 152, rect_scale=0.08, rect_scale_max=0.7)
    nd = forest.nodes
    c1 = (nd["r1"][:, 2].astype(np.int64) - nd["r1"][:, 0]) * (nd["r1"][:, 3].astype(np.int64) - nd["r1"][:, 1])
    c2 = (nd["r2"][:, 2].astype(np.int64) - nd["r2"][:, 0]) * (nd["r2"][:, 3].astype(np.int64) - nd["r2"][:, 1])
    assert (c1 != c2).mean() > 0.5
    cc = (np.maximum(c1, 1) * np.maximum(c2, 1)).astype(np.float64)
    thr = nd["threshold"]
    exact = np.round(thr * cc) / cc
    sel = np.arange(thr.size) % 5
    thr[sel == 0] = exact[sel == 0]
    thr[sel == 1] = np.nextafter(exact[sel == 1], np.inf)
    thr[sel == 2] = np.nextafter(exact[sel == 2], -np.inf)
    thr[sel == 3] = 0.0
    thr[11] = 65535.0; thr[12] = -65535.0; thr[13] = np.inf; thr[14] = -np.inf; thr[15] = np.nextafter(65535.0, 0.0)
    nd["r1"][20, 2] = nd["r1"][20, 0]                  # zero-width rectangle: c1 = 0
    nd["r2"][21, 3] = nd["r2"][21, 1]                  # zero-height rectangle: c2 = 0
    nd["r1"][22, 2] = nd["r1"][22, 0]; nd["r2"][22, 2] = nd["r2"][22, 0]      # both empty: 0 - 0 > thr
    model = synth.ModelParams(stepwidth=4)
    w, h = 240, 200
    frames = synth.biwi_batch(3, w, h, first=37)
    frames[1] = (frames[1] > 0) * 800                  # flat foreground: D = 0 for rectangles inside it
    frames[2] = np.where(frames[2] > 0, 65535, 0)      # saturated sums
    from test_gpu_parity import _check_frames
    if not int_test:
        os.environ["DH_NO_GENERAL_INT"] = "1"
    try:
        _check_frames(hp_mod, oracle, forest, model, frames.astype(np.uint16), synth.default_intrinsic(w, h), full=False)
    finally:
        os.environ.pop("DH_NO_GENERAL_INT", None)


def test_general_path_patches_beyond_255_keep_the_division_walk(hp_mod, oracle):
    """NodeG stores rectangle corners in bytes: a 300-pixel-wide patch takes the walk with the f64 divisions."""
    sw, sh, w, h = 300, 40, 640, 120
    forest = synth.synth_forest(4, 6, synth.FOREST_SEED_BASE + 153, patch=(sw, sh), rect_scale=0.1, rect_scale_max=0.5)
    model = synth.ModelParams(stepwidth=8, subimage_width=sw, subimage_height=sh)
    frames = np.stack([synth.biwi_like(640, 480, 5300 + i)[200:200 + h, :w] for i in range(2)]).copy()
    from test_gpu_parity import _check_frames
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), full=False)


@pytest.mark.parametrize("min_hits", [None, "1", "100000000"])
@pytest.mark.parametrize("leaf_hist", [True, False])
def test_first_regions_gathered_by_k_region(hp_mod, oracle, leaf_hist, min_hits):
    """Small batches with many hit records: k_region gathers the cells of both accumulators around the initial guesses with several
    workgroups per frame and k_cluster cuts its regions out of them (tests/test_gpu_round3.py has the windows that travel).  Forced on for every frame (DH_REGION_MIN_HITS=1), off (huge
    threshold) and automatic, with and without the leaf histogram: identical traces and poses (integer atomics are order-free)."""
    forest = synth.fit_forest(8, 10, synth.FOREST_SEED_BASE + 61, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=2)
    w, h = 320, 240
    frames = synth.biwi_batch(5, w, h, first=90)
    frames[4] = 0                                                     # a frame without any hit record
    env = {}
    if min_hits is not None:
        env["DH_REGION_MIN_HITS"] = min_hits
    if not leaf_hist:
        env["DH_NO_LEAF_HIST"] = "1"
    os.environ.update(env)
    try:
        from test_gpu_parity import _check_frames
        _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), full=False)
        rs = np.random.RandomState(4)
        _check_frames(hp_mod, oracle, forest, model, frames[:3], synth.default_intrinsic(w, h),
                      rs.uniform(-100, 900, (3, 3)).astype(np.float32), rs.uniform(-1, 1, (3, 3)), np.array([3, 1, 2], dtype=np.uint8), full=False)
        _product_mode_check(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h))
        # the blocks are zeroed per batch: several batches of other frames
        # through ONE predictor, every pose against the oracle's
        K = synth.default_intrinsic(w, h)
        other = synth.biwi_batch(5, w, h, first=140)
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            for batch in (frames, other, frames[:2], other[::-1].copy(), frames):
                got = hp.predict_batch(batch, hp_mod.IntrinsicMatrix(K))
                assert _poses_equal(got, oracle.predict_batch(forest, model, batch, K))
    finally:
        for k in env:
            os.environ.pop(k, None)


def test_graph_replays_on_a_large_workspace(hp_mod, oracle):
    """Found by tools/soak.py: a graph captured on a workspace sized for 512 frames zero-fills 22 MB of per-batch counters per
    replay; as a memset NODE that fill left part of the range untouched from the second replay on (stale window counts and
    histograms -> wrong poses).  The fill is a kernel node now.  Several replays on new contents, also with k_region."""
    torch = pytest.importorskip("torch")
    from depthhead_amd._lib import POSE_DTYPE
    forest = synth.fit_forest(10, 12, synth.FOREST_SEED_BASE + 71, n_frames=12, subset=2000)
    model = synth.ModelParams(stepwidth=4)
    w, h, n = 640, 480, 2
    K = synth.default_intrinsic(w, h)
    intr = hp_mod.IntrinsicMatrix(K)
    dev = torch.device("cuda:0")
    frames = synth.biwi_batch(8, w, h, first=120)
    frames[3] = 0
    ref = oracle.predict_batch(forest, model, frames, K)
    fr = torch.zeros((n, h, w), dtype=torch.int16, device=dev)
    out = torch.zeros(n * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    for env in ({}, {"DH_REGION_MIN_HITS": "1", "DH_NO_LEAF_HIST": "1"}):
        os.environ.update(env)
        try:
            with hp_mod.HoughPrediction(forest, model, device=0) as hp:
                hp.reserve(512, w, h)
                hp.graph_capture(fr.data_ptr(), n, w, h, intr, out.data_ptr())
                for a, b in ((0, 1), (2, 3), (3, 3), (4, 5), (0, 7), (6, 6)):
                    fr.copy_(torch.from_numpy(frames[[a, b]].view(np.int16)))
                    hp.graph_launch(st.cuda_stream)
                    st.synchronize()
                    poses = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
                    assert _poses_equal(poses, ref[[a, b]]), (a, b, env)
        finally:
            for k in env:
                os.environ.pop(k, None)


def test_error_behaviour_of_the_round2_entry_points(hp_mod, hip_lib):
    """Status codes, never a fault: empty and NULL arguments, frames smaller than the patch, an output buffer that is too
    small, sizes that do not fit -- for the entry points added in round 2."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    from depthhead_amd import biwi
    from depthhead_amd._lib import DepthheadError, pinned_empty
    forest = synth.synth_forest(3, 5, 31)
    model = synth.ModelParams(stepwidth=4)
    K = hp_mod.IntrinsicMatrix(synth.default_intrinsic(160, 120))
    frames = synth.biwi_batch(2, 160, 120)
    pay = [biwi.encode_depth(f) for f in frames]
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        assert hp.predict_batch_rle([], K).shape == (0,)                                   # empty batch
        assert hp.predict_parameter_from2dhough(np.zeros((0, 120, 160), np.uint16), K).shape == (0,)
        assert hp.build_hough_image(np.zeros((0, 120, 160), np.uint16), K).shape == (0, 120, 160)
        with pytest.raises(DepthheadError) as ei:
            hp.predict_parameter_from2dhough(np.zeros((1, 60, 160), np.uint16), K)        # shorter than the patch
        assert ei.value.code == -5
        with pytest.raises(DepthheadError) as ei:
            hp.predict_batch_rle([biwi.encode_depth(np.zeros((60, 160), np.uint16))], K)   # decodes to a frame shorter than the patch
        assert ei.value.code == -5
        dev = torch.zeros((2, 120, 160), dtype=torch.int16, device="cuda:0")
        with pytest.raises(DepthheadError) as ei:
            hp.decode_depth_device(pay, dev.data_ptr(), 2 * 120 * 160 - 1)                 # output one pixel short
        assert ei.value.code == -1
        huge = struct.pack("<II", 70000, 70000) + struct.pack("<II", 0, 0)
        with pytest.raises(DepthheadError) as ei:
            hp.decode_depth_device([huge])                                                  # 4.9 G pixels: refused, not allocated
        assert ei.value.code == -5
        zero = struct.pack("<II", 0, 17)
        with pytest.raises(DepthheadError):
            hp.decode_depth_device([zero])
        # raw NULLs through the C ABI
        assert hip_lib.dh_predict_batch_rle(hp._ph, None, None, 1, None, None, None, None, None) == -1
        assert hip_lib.dh_predict_from2dhough(hp._ph, None, 1, 160, 120, None, None) == -1
        assert hip_lib.dh_build_hough_image(None, None, 1, 160, 120, None, None) == -1
        assert hip_lib.dh_biwi_decode_depth_device(hp._ph, None, None, 1, None, C.c_size_t(0), None, None) == -1
        assert hip_lib.dh_host_alloc(C.c_size_t(16), None) == -1
        assert hip_lib.dh_host_free(None) == 0
        assert hip_lib.dh_predict_batch_rle(hp._ph, None, None, -3, None, None, None, None, None) == -1
        # and the predictor still works
        ok = hp.predict_batch_rle(pay, K)
        assert ok.tobytes() == hp.predict_batch(frames, K).tobytes()
    small = pinned_empty((0,), np.uint16)
    assert small.size == 0


import struct  # noqa: E402  (used by the test above)


@pytest.mark.parametrize("dense", [False, True])
def test_box_image_stays_exact_across_batches(hp_mod, oracle, dense):
    """k_boxsum skips stores of zero over zero, relying on what the same frame slot held after the previous batch.  One
    predictor, a sequence of batches built to break that: content appears, moves, vanishes and reappears in the same
    cells; batch sizes change (other band partition of the image, other frame slots); the sibling consumers run in
    between.  Leaf indices (taps on from the start, so no reallocation resets the state) and poses against the oracle."""
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 91, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=4)
    w, h = 640, 480
    K = synth.default_intrinsic(w, h)
    intr = hp_mod.IntrinsicMatrix(K)
    rs = np.random.RandomState(17)
    base = synth.biwi_batch(6, w, h, first=140)
    zero = np.zeros((h, w), np.uint16)
    dots = ((rs.rand(h, w) < 0.002) * rs.randint(1, 4000, (h, w))).astype(np.uint16)       # isolated pixels: thin non-zero stripes of sums
    full = rs.randint(500, 3000, (h, w)).astype(np.uint16)
    shifted = np.roll(base[0], (37, -91), axis=(0, 1))
    half = base[1].copy(); half[:, :320] = 0
    seq = [
        np.stack([base[0], base[1], base[2]]),
        np.stack([zero, base[1], zero]),                 # slots 0 and 2 lose their content
        np.stack([base[0], zero, dots]),                 # slot 0 regains exactly what it had; slot 1 loses; slot 2 gets sparse pixels
        np.stack([shifted]),                             # one frame: many short bands
        np.stack([full, half, base[3], base[4], base[5], zero, dots, base[0], shifted]),   # nine frames: other band partition, new slots
        np.stack([zero, zero, zero]),
        np.stack([half, full, base[2]]),
    ]
    if dense:
        os.environ["DH_BOX_DENSE"] = "1"
    try:
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            hp.debug_enable(True)
            hp.reserve(9, w, h)
            for bi, frames in enumerate(seq):
                n = frames.shape[0]
                poses = hp.predict_batch(frames, intr)
                leaf = hp.debug_leaf_indices(n, w, h)
                flags = hp.debug_patch_flags(n, w, h)
                for i in range(n):
                    ref = oracle.predict(forest, model, frames[i], K)
                    assert np.array_equal(flags[i], ref.patch_flags), (bi, i)
                    assert np.array_equal(leaf[i], ref.leaf_idx), (bi, i)
                    assert np.array_equal(poses["mid_point"][i], ref.mid_point) and np.array_equal(poses["rotation"][i], ref.rotation), (bi, i)
                if bi in (1, 4):                          # the sibling consumers fill the same images
                    m = hp.predict_mask(frames[:2])
                    assert np.array_equal(m[1], oracle.predict_mask(forest, model, frames[1]))
                    hp.debug_enable(True)
    finally:
        os.environ.pop("DH_BOX_DENSE", None)


def _stunted_forest(seed):
    """A forest whose trees end at every depth from 0 (a root that is a leaf) on, so that paths end inside the levels the
    uniform path walks from its LDS copy of the tree tops."""
    from depthhead_amd.forest import NODE_DTYPE, Forest
    base = synth.synth_forest(6, 7, seed)
    rs = np.random.RandomState(seed)
    nodes = base.nodes.copy()
    roots = base.roots.copy()
    # cut sub-trees short: point some children straight at a leaf
    for i in rs.choice(len(nodes), size=len(nodes) // 5, replace=False):
        nodes["child_one" if rs.rand() < 0.5 else "child_zero"][i] = ~int(rs.randint(0, len(base.leaf_prob)))
    roots[1] = ~int(rs.randint(0, len(base.leaf_prob)))                 # a tree that is one leaf
    roots[4] = int(nodes["child_zero"][roots[4]]) if nodes["child_zero"][roots[4]] >= 0 else roots[4]
    return Forest(roots, nodes, base.leaf_prob, base.off_begin, base.rot_begin, base.offsets, base.rotations)


@pytest.mark.parametrize("knob", [None, "DH_NO_ABSORB=1", "DH_TOP_LEVELS=0", "DH_TOP_LEVELS=1", "DH_TOP_LEVELS=3", "DH_TOP_LEVELS=8"])
def test_walk_table_variants(hp_mod, oracle, knob):
    """The uniform path's walks: the guarded node table (DH_NO_ABSORB) and the unguarded walk table with 0..8 tree levels taken
    from LDS, on a forest with paths ending at every depth (also a tree that is a single leaf): same leaves, votes and poses."""
    from test_gpu_parity import _check_frames
    forest = _stunted_forest(77)
    model = synth.ModelParams(stepwidth=4)
    w, h = 200, 168
    frames = synth.biwi_batch(3, w, h, first=33)
    env = dict([knob.split("=")]) if knob else {}
    os.environ.update(env)
    try:
        _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), full=True)
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            hp.reserve(3, w, h)
            geo = hp.debug_geometry()
            assert geo["uniform"] == 1
            assert geo["walk_table"] == (0 if knob == "DH_NO_ABSORB=1" else 1)
            if knob and knob.startswith("DH_TOP_LEVELS"):
                assert geo["top_levels"] == int(env["DH_TOP_LEVELS"])
    finally:
        for k in env:
            os.environ.pop(k, None)


@pytest.mark.parametrize("where", ["root", "deep", "many", "too many"])
def test_ambiguous_thresholds_in_the_walk_table(hp_mod, oracle, where):
    """A node whose threshold times the rectangle area is (within 2^-20 of) an integer has an ambiguity band that only the
    reference's f64 arithmetic decides (k_nodes_compact).  The walk table carries such a node as two entries and a walk
    that lands in the band is resolved and walks on -- at the root (inside the tree tops: the walk leaves the LDS heap there),
    deep in a tree, at hundreds of nodes; beyond DH_AMB_CAP of them the predictor keeps the guarded node table.  Flat frames
    make every rectangle difference 0, so thresholds of 0 put EVERY visit of such a node into its band."""
    from test_gpu_parity import _check_frames
    from depthhead_amd.forest import Forest
    big = where == "too many"
    forest = synth.synth_forest(6, 12 if big else 7, 91)
    nodes = forest.nodes.copy()
    rs = np.random.RandomState(5)
    if where == "root":
        pick = [int(r) for r in forest.roots[:3]]
    elif where == "deep":
        pick = [int(i) for i in rs.choice(len(nodes), 5, replace=False)]
    else:
        pick = list(range(len(nodes))) if big else [int(i) for i in rs.choice(len(nodes), min(300, len(nodes)), replace=False)]
    nodes["threshold"][pick] = np.where(rs.rand(len(pick)) < 0.5, 0.0, 3.0)
    forest = Forest(forest.roots, nodes, forest.leaf_prob, forest.off_begin, forest.rot_begin, forest.offsets, forest.rotations)
    assert not big or len(nodes) > 4096
    model = synth.ModelParams(stepwidth=4)
    w, h = 160, 120
    frames = synth.biwi_batch(3, w, h, first=5)
    frames[1] = 900                                                   # flat: every difference of two rectangle means is 0
    frames[2, :, 80:] = 1200
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), full=True)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.reserve(2, w, h)
        geo = hp.debug_geometry()
        assert geo["uniform"] == 1 and geo["walk_table"] == (0 if big else 1)


@pytest.mark.parametrize("size", [(640, 480), (320, 240), (200, 160), (330, 250)])
def test_vote_cells_at_grid_borders(hp_mod, oracle, size):
    """k_vote takes the cell of the 20 x 20 guess grid from an approximate quotient unless the quotient lies next to a cell
    border (frames whose sides are multiples of 20; 330 x 250 is not and always divides).  Here every vote lands ON or within
    1e-6 .. 1e-2 pixels of a border: a flat frame at z = fx makes the window centre (x - cx, y - cy, fx), so a vote with offset
    (ox, oy, 0) projects to exactly (x - ox, y - oy) (prediction.rs:647-676), and the offsets put that on multiples of the
    cell width / height for some windows and a hair beside them for the others.  A one-leaf forest: every window hits it."""
    from depthhead_amd.forest import NODE_DTYPE, Forest
    w, h = size
    K = synth.default_intrinsic(w, h)
    fx = float(K[0, 0])                                              # (an integer except at 330 x 250, where nothing is exact anyway)
    frames = np.full((2, h, w), int(round(fx)), dtype=np.uint16)
    frames[1, :, : w // 2] = 0                                       # second frame: half of it background
    eps = [0.0, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 3e-4, -3e-4, 1e-3, -1e-3, 1e-2]
    cw, ch = w / 20.0, h / 20.0
    # window centres sit at 40 + 4 i: offsets that move them onto / beside multiples of the cell size
    offs = np.array([[(40 % cw) + 4 * (j % 3) + e, (40 % ch) + 4 * (j % 2) - e, 0.0] for j, e in enumerate(eps)], dtype=np.float32)
    rots = np.tile(np.array([[0.1, -0.2, 0.3]]), (2, 1))
    forest = Forest(np.array([~0], dtype=np.int32), np.zeros(0, dtype=NODE_DTYPE), np.array([1.0]),
                    np.array([0, len(offs)], dtype=np.uint32), np.array([0, len(rots)], dtype=np.uint32), offs, rots)
    model = synth.ModelParams(stepwidth=4)
    from test_gpu_parity import _check_frames
    _check_frames(hp_mod, oracle, forest, model, frames, K, full=True)
    # the same through a general (non-pinhole) intrinsic matrix: the cell test does not depend on the projection's form
    K2 = K.copy()
    K2[0, 1] = 0.25
    K2[2, 0] = 1e-5
    _check_frames(hp_mod, oracle, forest, model, frames, K2, full=False)


def test_forest_of_very_many_trees_keeps_the_guarded_walks(hp_mod, oracle):
    """The walk table costs 12 bytes of LDS per tree; a forest of 700 (tiny) trees walks the guarded node table instead."""
    from test_gpu_parity import _check_frames
    forest = synth.synth_forest(700, 2, 17)
    model = synth.ModelParams(stepwidth=8)
    w, h = 160, 120
    frames = synth.biwi_batch(1, w, h, first=9)
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), full=False)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.reserve(1, w, h)
        geo = hp.debug_geometry()
        assert geo["uniform"] == 1 and geo["walk_table"] == 0
    forest = synth.synth_forest(600, 2, 17)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.reserve(1, w, h)
        assert hp.debug_geometry()["walk_table"] == 1


@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("knob", [None, "DH_NO_TILE_LIST"])
def test_flagged_tile_lists_in_product_mode(hp_mod, oracle, general, knob):
    """Product mode hands k_traverse the flagged tiles as compact lists (k_tile_list), one per frame mod 8: frames that are
    empty, all noise (every tile flagged), a blob in one corner, ordinary subjects; 11 and 3 frames (not multiples of 8);
    uniform and general path; and the same with a workgroup per tile position (DH_NO_TILE_LIST)."""
    forest = synth.fit_forest(6, 9, synth.FOREST_SEED_BASE + 83, n_frames=10, subset=1500)
    model = synth.ModelParams(stepwidth=4)
    w, h = 320, 240
    frames = synth.biwi_batch(11, w, h, first=600)
    rs = np.random.RandomState(8)
    frames[1] = 0
    frames[4] = rs.randint(400, 1400, (h, w)).astype(np.uint16)
    frames[6] = 0
    frames[6, 200:, 280:] = 900
    frames[9] = 0
    frames[10, :, :160] = 0
    env = {}
    if general:
        env["DH_FORCE_GENERAL"] = "1"
    if knob:
        env[knob] = "1"
    _product_mode_check(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), env=env)
    _product_mode_check(hp_mod, oracle, forest, model, frames[3:6], synth.default_intrinsic(w, h), env=env)
