"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, stage by stage.

Bar (BASELINE.json north_star): leaf indices bit-exact; mid_point / rotation within 1e-4 -- both are
integer-grid values (src/hough/prediction.rs:477-489), so the tests demand exact equality, which
implies the 1e-4 tolerance.  Integer stages (flags, grids, guesses, accumulators, mean-shift
positions) are compared bit for bit.
"""
import numpy as np
import pytest

from depthhead_amd import synth
from depthhead_amd.forest import Forest, NODE_DTYPE

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4   # north_star tolerance; results are integer-grid so we also assert equality


@pytest.fixture(scope="module")
def hp_mod(hip_lib):
    from depthhead_amd import prediction
    return prediction


def _check_frames(hp_mod, oracle, forest, model, frames, K, midp=None, rot=None, mask=None, full=True):
    n, h, w = frames.shape
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.debug_enable(True)
        poses = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K), midp, rot, mask)
        leaf = hp.debug_leaf_indices(n, w, h)
        flags = hp.debug_patch_flags(n, w, h)
        pos_grid, rot_grid = hp.debug_grids(n)
        guesses = hp.debug_guesses(n)
        tr_mid, st_mid = hp.debug_meanshift(n, 0)
        tr_rot, st_rot = hp.debug_meanshift(n, 1)
        votes = [(hp.debug_votes(i, 0), hp.debug_votes(i, 1)) for i in range(n)] if full else None
    it = model.meanshift_iterations
    for i in range(n):
        mg = None if midp is None or (mask is not None and not mask[i] & 1) else midp[i]
        rg = None if rot is None or (mask is not None and not mask[i] & 2) else rot[i]
        ref = oracle.predict(forest, model, frames[i], K, mg, rg)
        assert np.array_equal(leaf[i], ref.leaf_idx), f"frame {i}: leaf indices"
        assert np.array_equal(flags[i], ref.patch_flags), f"frame {i}: patch flags"
        assert np.array_equal(pos_grid[i], ref.pos_grid), f"frame {i}: 20x20 guess grid"
        assert np.array_equal(rot_grid[i], ref.rot_grid), f"frame {i}: 20^3 guess grid"
        assert np.array_equal(guesses[i, :3], ref.guess_mid), f"frame {i}: mid guess {guesses[i]} vs {ref.guess_mid}"
        assert np.array_equal(guesses[i, 3:], ref.guess_rot), f"frame {i}: rot guess {guesses[i]} vs {ref.guess_rot}"
        if full:
            assert np.array_equal(hp_mod.aggregate_votes(votes[i][0]), ref.mid_cells), f"frame {i}: mid accumulator"
            assert np.array_equal(hp_mod.aggregate_votes(votes[i][1]), ref.rot_cells), f"frame {i}: rot accumulator"
        for name, tr, st, rtr in (("mid", tr_mid, st_mid, ref.ms_trace_mid), ("rot", tr_rot, st_rot, ref.ms_trace_rot)):
            k = rtr.shape[0]   # oracle steps + 1
            # the kernel stops at a fixed point and reports the remaining (identical) steps as done
            assert st[i] + 1 >= k or st[i] == it, f"frame {i}: {name} steps {st[i]} vs {k - 1}"
            assert np.array_equal(tr[i, :k], rtr), f"frame {i}: {name} mean-shift trace"
        assert np.array_equal(poses["mid_point"][i], ref.mid_point), f"frame {i}: {poses['mid_point'][i]} vs {ref.mid_point}"
        assert np.array_equal(poses["rotation"][i], ref.rotation), f"frame {i}: {poses['rotation'][i]} vs {ref.rotation}"
        assert np.max(np.abs(poses["mid_point"][i] - ref.mid_point)) <= POSE_TOL
        assert np.max(np.abs(poses["rotation"][i] - ref.rotation)) <= POSE_TOL
    return poses


class general_path:
    """Force the general (SAT + f64 division) traversal instead of the uniform-rectangle fast path."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        import os
        if self.on:
            os.environ["DH_FORCE_GENERAL"] = "1"

    def __exit__(self, *exc):
        import os
        os.environ.pop("DH_FORCE_GENERAL", None)


@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("w,h,step,trees,depth", [
    (160, 120, 4, 5, 8),
    (200, 152, 3, 10, 12),
    (320, 240, 1, 3, 6),     # BASELINE config 5 geometry, small forest
    (320, 240, 7, 10, 15),
])
def test_stagewise_small(hp_mod, oracle, w, h, step, trees, depth, general):
    forest = synth.synth_forest(trees, depth, synth.FOREST_SEED_BASE + 100 + step)
    with general_path(general):
        model = synth.ModelParams(stepwidth=step)
        frames = synth.biwi_batch(3, w, h, first=10)
        _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h))


def test_mixed_rectangle_sizes(hp_mod, oracle):
    """Rectangles of different sizes in one forest (c1 != c2): the general traversal."""
    forest = synth.synth_forest(8, 11, synth.FOREST_SEED_BASE + 150, rect_scale=0.1, rect_scale_max=0.6)
    sizes = {(int(r[2] - r[0]), int(r[3] - r[1])) for r in forest.nodes["r1"][:50]}
    assert len(sizes) > 5
    model = synth.ModelParams(stepwidth=4)
    frames = synth.biwi_batch(3, 320, 240, first=15)
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(320, 240))


def test_integer_threshold_edge_cases(hp_mod, oracle):
    """Thresholds that sit exactly on, or within one ulp of, a representable box-sum difference:
    the integer test must hand those to the exact f64 path (k_nodes_compact's ambiguity band)."""
    forest = synth.synth_forest(6, 9, synth.FOREST_SEED_BASE + 151)
    c = 24.0 * 24.0
    thr = forest.nodes["threshold"]
    k = np.round(thr * c)
    exact = k / c                                   # delta == thr is reachable: d > thr must be false there
    sel = np.arange(thr.size) % 4
    thr[sel == 0] = exact[sel == 0]
    thr[sel == 1] = np.nextafter(exact[sel == 1], np.inf)
    thr[sel == 2] = np.nextafter(exact[sel == 2], -np.inf)
    thr[sel == 3] = np.where(np.arange(thr.size)[sel == 3] % 8 == 3, 0.0, thr[sel == 3])
    thr[5] = 1e9; thr[6] = -1e9; thr[7] = 65535.0; thr[8] = -65535.0; thr[9] = np.inf; thr[10] = -np.inf
    model = synth.ModelParams(stepwidth=4)
    w, h = 240, 200
    frames = synth.biwi_batch(3, w, h, first=33)
    frames[1] = (frames[1] > 0) * 800               # flat foreground: many exactly-equal box sums (delta == 0)
    frames[2] = np.where(frames[2] > 0, 65535, 0)   # saturated: largest possible sums
    _check_frames(hp_mod, oracle, forest, model, frames.astype(np.uint16), synth.default_intrinsic(w, h))
    with general_path(True):
        _check_frames(hp_mod, oracle, forest, model, frames.astype(np.uint16), synth.default_intrinsic(w, h))




def test_config1_single_frame_stride10(hp_mod, oracle):
    """BASELINE config 1: one 640x480 frame, 10-tree forest, trainer stride 10."""
    forest = synth.synth_forest(10, 15, synth.FOREST_SEED_BASE + 1)
    model = synth.ModelParams(stepwidth=10)
    frames = synth.biwi_batch(1, 640, 480)
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic())


def test_config2_stride4_batch(hp_mod, oracle):
    """BASELINE config 2 geometry (640x480, 10 trees, stride 4) on a 6-frame slice."""
    forest = synth.synth_forest(10, 15, synth.FOREST_SEED_BASE + 2)
    model = synth.ModelParams(stepwidth=4)
    frames = synth.biwi_batch(6, 640, 480, first=3)
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic())


def test_config3_deep_forest_stride2(hp_mod, oracle):
    """BASELINE config 3 geometry (stride 2, depth 20) with 12 trees to keep the oracle quick."""
    forest = synth.synth_forest(12, 20, synth.FOREST_SEED_BASE + 3)
    model = synth.ModelParams(stepwidth=2)
    frames = synth.biwi_batch(2, 640, 480, first=40)
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(), full=False)


def test_guess_overrides(hp_mod, oracle):
    """midp_guess / rot_guess Options (prediction.rs:437-460), per frame via guess_mask."""
    forest = synth.synth_forest(8, 10, synth.FOREST_SEED_BASE + 4)
    model = synth.ModelParams(stepwidth=5)
    w, h, n = 320, 240, 4
    frames = synth.biwi_batch(n, w, h, first=20)
    K = synth.default_intrinsic(w, h)
    base = _check_frames(hp_mod, oracle, forest, model, frames, K)
    midp = (base["mid_point"] + np.array([3.7, -2.2, 5.9], dtype=np.float32)).astype(np.float32)
    rot = base["rotation"] + 0.04
    _check_frames(hp_mod, oracle, forest, model, frames, K, midp, rot)
    mask = np.array([0, 1, 2, 3], dtype=np.uint8)
    _check_frames(hp_mod, oracle, forest, model, frames, K, midp, rot, mask)
    wild_m = np.array([[1e20, -1e20, np.nan]] * n, dtype=np.float32)   # saturating / NaN casts
    wild_r = np.array([[1e300, -7.0, np.nan]] * n, dtype=np.float64)
    _check_frames(hp_mod, oracle, forest, model, frames, K, wild_m, wild_r)


def test_edge_frames(hp_mod, oracle):
    """All-background frame, frame exactly one patch wide (zero window positions), saturated frame."""
    forest = synth.synth_forest(4, 8, synth.FOREST_SEED_BASE + 5)
    model = synth.ModelParams(stepwidth=4)
    K = synth.default_intrinsic(160, 120)
    zero = np.zeros((2, 120, 160), dtype=np.uint16)
    zero[1, 60, 80] = 1   # a single non-zero pixel
    _check_frames(hp_mod, oracle, forest, model, zero, K)
    full = np.full((1, 120, 160), 65535, dtype=np.uint16)
    _check_frames(hp_mod, oracle, forest, model, full, K)
    tiny = synth.biwi_batch(1, 160, 120)[:, :80, :80].copy()   # w == sw: the loops run zero times
    _check_frames(hp_mod, oracle, forest, model, tiny, synth.default_intrinsic(80, 80))
    one = synth.biwi_batch(1, 160, 120)[:, :81, :83].copy()     # exactly one row of three positions... (1 x 3)
    _check_frames(hp_mod, oracle, forest, synth.ModelParams(stepwidth=1), one, synth.default_intrinsic(83, 81))


def _hand_forest():
    """Two trees built by hand: a single-leaf tree, zero-area rectangles, a one-vote leaf (its
    covariance is 0/0 = NaN so it never votes, meancov_estimation.rs:376), a leaf whose rotation
    wraps at +-180 degrees, equal-threshold ties."""
    nodes = np.zeros(3, dtype=NODE_DTYPE)
    nodes[0] = ((10, 10, 34, 34), (40, 40, 64, 64), 0.0, 1, ~0)          # avg1-avg2 > 0 ? leaf0 : node1
    nodes[1] = ((0, 0, 0, 0), (5, 5, 5, 30), -1.0, ~1, 2)                # two empty rects: 0-0 > -1 -> node2
    nodes[2] = ((0, 0, 80, 80), (0, 0, 1, 1), 100.0, ~2, ~3)
    roots = np.array([0, ~4], dtype=np.int32)                            # tree 1 is a single leaf
    prob = np.array([0.9, 0.5, 1.0, 0.8, 1.0])
    noff = [3, 1, 4, 2, 3]
    off_begin = np.concatenate([[0], np.cumsum(noff)]).astype(np.uint32)
    offsets = np.array([[10, 5, -20], [12, 6, -22], [9, 4, -19],
                        [0, 0, 0],
                        [-30, 10, 15], [-31, 11, 16], [-29, 9, 14], [-30.5, 10.5, 15.5],
                        [1e6, 0, 0], [-1e6, 0, 0],                        # huge spread: offset gate fails
                        [0.5, 0.5, 2000.0], [0.25, 0.75, 1999.0], [0.1, 0.2, 2001.0]], dtype=np.float32)  # np.z < 0 skips
    rotations = np.array([[181.0, -185.0, 10], [178.0, -178.0, 11], [179.9, -179.9, 9],   # 181 -> bin 120 -> 0; -185 -> -1 -> 119
                          [5, 5, 5],
                          [20, -30, 40], [21, -31, 41], [19, -29, 39], [20.5, -30.5, 40.5],
                          [0, 0, 0], [90, 90, 90],                        # trace > 400: rotation gate fails
                          [-2.9, 2.9, 0.0], [-3.1, 3.1, -0.0], [-1.0, 1.0, 0.5]], dtype=np.float64)
    return Forest(roots, nodes, prob, off_begin, off_begin.copy(), offsets, rotations)


def test_hand_built_forest(hp_mod, oracle):
    forest = _hand_forest()
    model = synth.ModelParams(stepwidth=6, meanshift_iterations=7)
    frames = synth.biwi_batch(3, 200, 160, first=5)
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(200, 160))


def test_non_square_odd_patch_and_dense_intrinsic(hp_mod, oracle):
    """Odd, non-square patch (left/right halves differ, prediction.rs:535-538) and the dense
    intrinsic of the reference's own test (types.rs:478)."""
    forest = synth.synth_forest(5, 9, synth.FOREST_SEED_BASE + 6, patch=(61, 47))
    model = synth.ModelParams(stepwidth=3, subimage_width=61, subimage_height=47, gaussian_sigma=3.5, meanshift_iterations=5)
    frames = synth.biwi_batch(2, 180, 140, first=7)
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(180, 140))
    K = np.array([[22.0, 11.4, 12.11], [2.1, 4.1, 2.11], [1.3, 3.1, 19.0]], dtype=np.float32)
    _check_frames(hp_mod, oracle, forest, model, frames, K)


def test_update_sigma(hp_mod, oracle):
    forest = synth.synth_forest(6, 10, synth.FOREST_SEED_BASE + 7)
    model = synth.ModelParams(stepwidth=4)
    frames = synth.biwi_batch(2, 320, 240, first=30)
    K = synth.default_intrinsic(320, 240)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        a = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        hp.update_sigma(-1.0)            # ignored (prediction.rs:321)
        assert hp.sigma() == 8.0
        hp.update_sigma(2.5)
        assert hp.sigma() == 2.5
        b = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
    m2 = synth.ModelParams(stepwidth=4, gaussian_sigma=2.5)
    for i in range(2):
        ra = oracle.predict(forest, model, frames[i], K, taps=False)
        rb = oracle.predict(forest, m2, frames[i], K, taps=False)
        assert np.array_equal(a["mid_point"][i], ra.mid_point) and np.array_equal(a["rotation"][i], ra.rotation)
        assert np.array_equal(b["mid_point"][i], rb.mid_point) and np.array_equal(b["rotation"][i], rb.rotation)


def test_single_frame_api_matches_reference_signature(hp_mod, oracle):
    """predict_parameter_parallel(img, intrinsic, midp_guess, rot_guess) -> PredictionResult."""
    forest = synth.synth_forest(10, 15, synth.FOREST_SEED_BASE + 1)
    model = synth.ModelParams(stepwidth=10)
    img = synth.biwi_like(640, 480, synth.FRAME_SEED_BASE + 77)
    intr = hp_mod.IntrinsicMatrix.default_kinect_intrinsic()
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        r0 = hp.predict_parameter_parallel(img, intr, None, None)
        r1 = hp.predict_parameter(img, intr, None, None)
        r2 = hp.predict_parameter_parallel(img, intr, r0.mid_point, r0.rotation)   # live_prediction.rs:81-86
    ref0 = oracle.predict(forest, model, img, intr.mat, taps=False)
    ref2 = oracle.predict(forest, model, img, intr.mat, ref0.mid_point, ref0.rotation, taps=False)
    assert np.array_equal(r0.mid_point, ref0.mid_point) and np.array_equal(r0.rotation, ref0.rotation)
    assert np.array_equal(r1.mid_point, r0.mid_point) and np.array_equal(r1.rotation, r0.rotation)
    assert np.array_equal(r2.mid_point, ref2.mid_point) and np.array_equal(r2.rotation, ref2.rotation)
    assert r0.bounding_box == (0, 0, 0, 0)


def test_full_size_batch_properties(hp_mod, oracle):
    """BASELINE config 2 at full size (256 frames): size-independent properties -- results do not
    depend on batch composition or position, runs are deterministic, and a sample of frames matches
    the oracle."""
    forest = synth.synth_forest(10, 15, synth.FOREST_SEED_BASE + 2)
    model = synth.ModelParams(stepwidth=4)
    base = synth.biwi_batch(32, 640, 480)
    frames = np.concatenate([base] * 8)                   # 256 frames, every frame appears 8 times
    perm = np.random.RandomState(1).permutation(256)
    K = synth.default_intrinsic()
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        a = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        b = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
        c = hp.predict_batch(frames[perm], hp_mod.IntrinsicMatrix(K))
        d = hp.predict_batch(frames[:5], hp_mod.IntrinsicMatrix(K))
    assert a.tobytes() == b.tobytes()                      # deterministic
    assert np.array_equal(c["mid_point"], a["mid_point"][perm]) and np.array_equal(c["rotation"], a["rotation"][perm])
    assert np.array_equal(d["mid_point"], a["mid_point"][:5])
    for r in range(1, 8):                                  # replicas agree
        assert np.array_equal(a["mid_point"][r * 32:(r + 1) * 32], a["mid_point"][:32])
        assert np.array_equal(a["rotation"][r * 32:(r + 1) * 32], a["rotation"][:32])
    ref = oracle.predict_batch(forest, model, base[:16], K)
    assert np.array_equal(a["mid_point"][:16], ref["mid_point"]) and np.array_equal(a["rotation"][:16], ref["rotation"])


def test_device_resident_api_with_torch_stream(hp_mod, oracle):
    """dh_predict_batch_device on torch-owned device memory and a non-default torch stream."""
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    forest = synth.synth_forest(6, 10, synth.FOREST_SEED_BASE + 8)
    model = synth.ModelParams(stepwidth=4)
    w, h, n = 320, 240, 5
    frames = synth.biwi_batch(n, w, h, first=60)
    K = synth.default_intrinsic(w, h)
    dev = torch.device("cuda:0")
    from depthhead_amd._lib import POSE_DTYPE
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.reserve(n, w, h)
        fr = torch.from_numpy(frames.view(np.int16)).to(dev)
        out = torch.zeros(n * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            hp.predict_batch_device(fr.data_ptr(), n, w, h, hp_mod.IntrinsicMatrix(K), out.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        poses = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
    ref = oracle.predict_batch(forest, model, frames, K)
    assert np.array_equal(poses["mid_point"], ref["mid_point"]) and np.array_equal(poses["rotation"], ref["rotation"])


@pytest.mark.parametrize("general", [False, True])
def test_device_frames_at_a_two_byte_aligned_address(hp_mod, oracle, general):
    """Device-resident frames that start on an odd 2-byte boundary (a slice of a larger buffer): the kernels'
    8-byte row loads must fall back to narrow loads."""
    torch = pytest.importorskip("torch")
    forest = synth.synth_forest(5, 9, synth.FOREST_SEED_BASE + 8)
    model = synth.ModelParams(stepwidth=4)
    w, h, n = 320, 240, 3
    frames = synth.biwi_batch(n, w, h, first=60)
    K = synth.default_intrinsic(w, h)
    dev = torch.device("cuda:0")
    from depthhead_amd._lib import POSE_DTYPE
    with general_path(general):
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            base = torch.zeros(n * w * h + 1, dtype=torch.int16, device=dev)
            base[1:] = torch.from_numpy(frames.view(np.int16).reshape(-1)).to(dev)
            fr = base[1:]
            assert fr.data_ptr() % 4 == 2
            out = torch.zeros(n * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            hp.predict_batch_device(fr.data_ptr(), n, w, h, hp_mod.IntrinsicMatrix(K), out.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
            torch.cuda.synchronize(dev)
            poses = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
    ref = oracle.predict_batch(forest, model, frames, K)
    assert np.array_equal(poses["mid_point"], ref["mid_point"]) and np.array_equal(poses["rotation"], ref["rotation"])


def test_golden_fixtures_without_oracle(hp_mod):
    """HIP path against the committed fixtures (tests/golden/*.npz) -- no oracle involved."""
    import golden_util
    for name in golden_util.names():
        forest, model, frames, K, exp = golden_util.load(name)
        n, h, w = frames.shape
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            hp.debug_enable(True)
            poses = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))
            leaf, flags = hp.debug_leaf_indices(n, w, h), hp.debug_patch_flags(n, w, h)
            pos_grid, rot_grid = hp.debug_grids(n)
            guesses = hp.debug_guesses(n)
            votes = [(hp.debug_votes(i, 0), hp.debug_votes(i, 1)) for i in range(n)]
            tr_mid, _ = hp.debug_meanshift(n, 0)
            tr_rot, _ = hp.debug_meanshift(n, 1)
        for i, e in enumerate(exp):
            assert np.array_equal(leaf[i], e["leaf_idx"]) and np.array_equal(flags[i], e["patch_flags"]), (name, i)
            assert np.array_equal(pos_grid[i], e["pos_grid"]) and np.array_equal(rot_grid[i], e["rot_grid"]), (name, i)
            assert np.array_equal(guesses[i], e["guess"]), (name, i)
            assert np.array_equal(hp_mod.aggregate_votes(votes[i][0]), e["mid_cells"]), (name, i)
            assert np.array_equal(hp_mod.aggregate_votes(votes[i][1]), e["rot_cells"]), (name, i)
            assert np.array_equal(tr_mid[i, :len(e["ms_trace_mid"])], e["ms_trace_mid"]), (name, i)
            assert np.array_equal(tr_rot[i, :len(e["ms_trace_rot"])], e["ms_trace_rot"]), (name, i)
            assert np.array_equal(poses["mid_point"][i], e["mid_point"]) and np.array_equal(poses["rotation"][i], e["rotation"]), (name, i)


def test_hand_verified_case(hp_mod):
    """The paper-checked case of tests/test_hand_case.py through the HIP path."""
    import test_hand_case as hc
    forest, model, img, K = hc.hand_case()
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        r = hp.predict_parameter_parallel(img, hp_mod.IntrinsicMatrix(K))
    hc.check(r.mid_point, r.rotation)


@pytest.mark.parametrize("trees", [64, 70])
def test_many_trees(hp_mod, oracle, trees):
    """64 / 70 shallow trees: several passes of walks per tile; k_emit gathers the leaves in batches of
    16 trees and, beyond 64 trees (no 64-bit voting mask), deals hit records with its generic search."""
    forest = synth.synth_forest(trees, 4, synth.FOREST_SEED_BASE + 160, full_depth=3)
    model = synth.ModelParams(stepwidth=4)
    frames = synth.biwi_batch(2, 240, 200, first=70)
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(240, 200), full=False)


class env_override:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        import os
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *exc):
        import os
        for k in self.kv:
            os.environ.pop(k, None)


@pytest.mark.parametrize("w,h,step,rect_scale,tile", [
    (642, 300, 4, 0.3, None),     # w % 4 != 0: 2-byte row loads in k_boxsum
    (640, 300, 4, 0.27, None),    # 21 x 21 rectangles: rw % 4 != 0 (unaligned LDS read-back), LDS ring
    (640, 300, 4, 0.4, None),     # 32 x 32 rectangles: rh - 1 > 28, no LDS ring
    (640, 300, 8, 0.3, None),     # stride 8: columns de-interleaved by 8
    (640, 300, 6, 0.3, None),     # stride 6: by 2
    (640, 300, 3, 0.3, None),     # odd stride: linear region
    (640, 300, 4, 0.3, "5,3"),    # forced tile with px % 4 != 0: element-wise region copy instead of direct-to-LDS
    (640, 300, 4, 0.3, "12,7"),
])
def test_uniform_path_variants(hp_mod, oracle, w, h, step, rect_scale, tile):
    """Every instance of k_boxsum (aligned / unaligned loads, ring / re-read) and every region layout of
    k_traverse's uniform path on frames wide enough for several tiles per row."""
    forest = synth.synth_forest(6, 9, synth.FOREST_SEED_BASE + 300 + step, rect_scale=rect_scale)
    model = synth.ModelParams(stepwidth=step)
    frames = np.stack([synth.biwi_like(644, 480, 3100 + i)[:h, :w] for i in range(2)]).copy()
    frames[1, :, : w // 2] = 0                      # half-empty frame: tiles skipped by their flags next to active ones
    with env_override(**({"DH_TILE": tile} if tile else {})):
        _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), full=False)
    with env_override(DH_BOX_NO_RING=1, **({"DH_TILE": tile} if tile else {})):
        if tile is None and step == 4 and rect_scale == 0.3:
            _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), full=False)


@pytest.mark.parametrize("general", [False, True])
def test_predict_mask_and_hough_votes(hp_mod, oracle, general):
    """predict_mask (prediction.rs:850-905) and the voting stage of
    build_hough_image (:760-840) reuse the walk kernel; both are byte-exact against the oracle."""
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 170, n_frames=12, subset=1500)
    assert (forest.leaf_prob >= 0.95).any()
    cases = [(320, 240, 4), (200, 160, 7), (168, 128, 1), (240, 200, 10)]
    with general_path(general):
        for w, h, step in cases:
            model = synth.ModelParams(stepwidth=step)
            frames = synth.biwi_batch(3, w, h, first=80)
            frames[2, : h // 2] = 0                                   # half the frame background
            K = synth.default_intrinsic(w, h)
            with hp_mod.HoughPrediction(forest, model, device=0) as hp:
                masks = hp.predict_mask(frames)
                votes = hp.build_hough_votes(frames, hp_mod.IntrinsicMatrix(K))
                poses = hp.predict_batch(frames, hp_mod.IntrinsicMatrix(K))     # the main path still works afterwards
                one = hp.predict_mask(frames[1])
            for i in range(3):
                assert np.array_equal(masks[i], oracle.predict_mask(forest, model, frames[i])), (w, h, step, i)
                assert np.array_equal(votes[i], oracle.hough_image(forest, model, frames[i], K)), (w, h, step, i)
            assert np.array_equal(one, masks[1])
            assert masks.max() > 0 and votes.max() > 0
            ref = oracle.predict_batch(forest, model, frames, K)
            assert np.array_equal(poses["mid_point"], ref["mid_point"]) and np.array_equal(poses["rotation"], ref["rotation"])


def test_hipgraph_replay(hp_mod, oracle):
    """BASELINE config 5's launch-bound regime: one captured batch replayed on new frame contents."""
    torch = pytest.importorskip("torch")
    from depthhead_amd._lib import POSE_DTYPE
    forest = synth.synth_forest(6, 10, synth.FOREST_SEED_BASE + 8)
    model = synth.ModelParams(stepwidth=1)
    w, h, n = 320, 240, 2
    K = synth.default_intrinsic(w, h)
    dev = torch.device("cuda:0")
    a, b = synth.biwi_batch(n, w, h, first=90), synth.biwi_batch(n, w, h, first=95)
    fr = torch.from_numpy(a.view(np.int16)).to(dev)
    out = torch.zeros(n * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        hp.graph_capture(fr.data_ptr(), n, w, h, hp_mod.IntrinsicMatrix(K), out.data_ptr())
        st = torch.cuda.current_stream(dev)
        for frames in (a, b, a):
            fr.copy_(torch.from_numpy(frames.view(np.int16)))
            hp.graph_launch(st.cuda_stream)
            st.synchronize()
            poses = np.frombuffer(out.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
            ref = oracle.predict_batch(forest, model, frames, K)
            assert np.array_equal(poses["mid_point"], ref["mid_point"]) and np.array_equal(poses["rotation"], ref["rotation"])


def test_batches_larger_than_the_resident_slice(hp_mod, oracle, monkeypatch):
    """Batches beyond DH_MAX_RESIDENT_FRAMES are walked slice by slice on the same workspace."""
    import depthhead_amd._lib as L
    forest = synth.synth_forest(5, 8, synth.FOREST_SEED_BASE + 180)
    model = synth.ModelParams(stepwidth=6)
    w, h, n = 200, 160, 11
    frames = synth.biwi_batch(n, w, h, first=130)
    K = synth.default_intrinsic(w, h)
    ref = oracle.predict_batch(forest, model, frames, K)
    # the limit is read once per process: drive the C entry point of a child process with a small slice
    import subprocess, sys, os, json, tempfile
    np.save(os.path.join(tempfile.gettempdir(), "dh_slice_frames.npy"), frames)
    code = (
        "import numpy as np, os, tempfile, json\n"
        "from depthhead_amd import synth\n"
        "from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix\n"
        "frames = np.load(os.path.join(tempfile.gettempdir(), 'dh_slice_frames.npy'))\n"
        f"forest = synth.synth_forest(5, 8, {synth.FOREST_SEED_BASE + 180}); model = synth.ModelParams(stepwidth=6)\n"
        f"K = synth.default_intrinsic({w}, {h})\n"
        "with HoughPrediction(forest, model) as hp:\n"
        "    p = hp.predict_batch(frames, IntrinsicMatrix(K)); m = hp.predict_mask(frames)\n"
        "print(json.dumps({'mid': p['mid_point'].tolist(), 'rot': p['rotation'].tolist(), 'mask': int(m.astype(np.int64).sum())}))\n")
    env = dict(os.environ, DH_MAX_RESIDENT_FRAMES="4")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert np.array_equal(np.array(got["mid"], dtype=np.float32), ref["mid_point"]) and np.array_equal(np.array(got["rot"]), ref["rotation"])
    assert got["mask"] == sum(int(oracle.predict_mask(forest, model, f).astype(np.int64).sum()) for f in frames)


def test_randomized_differential(hp_mod, oracle):
    """40 seeded random configurations (frame size, patch size, stride, forest shape, rectangle mix,
    sigma, iterations, intrinsics, guesses): HIP path == oracle on every pose and on leaf indices."""
    rs = np.random.RandomState(20260104)
    for case in range(40):
        sw, sh = int(rs.randint(24, 97)), int(rs.randint(24, 97))
        w, h = sw + int(rs.randint(0, 140)), sh + int(rs.randint(0, 110))
        step = int(rs.randint(1, 12))
        trees, depth = int(rs.randint(1, 13)), int(rs.randint(1, 11))
        mixed = rs.rand() < 0.4
        forest = synth.synth_forest(trees, depth, 5000 + case, patch=(sw, sh), rect_scale=float(rs.uniform(0.08, 0.5)),
                                    rect_scale_max=float(rs.uniform(0.5, 0.9)) if mixed else None,
                                    full_depth=int(rs.randint(0, depth + 1)), p_split=float(rs.uniform(0.4, 0.95)))
        model = synth.ModelParams(stepwidth=step, subimage_width=sw, subimage_height=sh,
                                  gaussian_sigma=float(rs.uniform(0.5, 30.0)), meanshift_iterations=int(rs.randint(0, 25)))
        n = int(rs.randint(1, 4))
        frames = np.stack([synth.biwi_like(max(w, 96), max(h, 96), 9000 + case * 7 + i)[:h, :w] for i in range(n)]).copy()
        if rs.rand() < 0.3:
            frames[0] = (rs.rand(h, w) < 0.02) * rs.randint(1, 65536, (h, w))        # sparse speckle, full u16 range
        f = float(rs.uniform(200, 900))
        K = np.array([[f, rs.uniform(-2, 2), w / 2 + rs.uniform(-20, 20)], [0, f * rs.uniform(0.9, 1.1), h / 2], [0, 0, 1]], dtype=np.float32)
        midp = rs.uniform(-300, 1500, (n, 3)).astype(np.float32) if rs.rand() < 0.3 else None
        rot = rs.uniform(-1.5, 1.5, (n, 3)) if rs.rand() < 0.3 else None
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            hp.debug_enable(True)
            poses = hp.predict_batch(frames.astype(np.uint16), hp_mod.IntrinsicMatrix(K), midp, rot)
            leaf = hp.debug_leaf_indices(n, w, h)
        for i in range(n):
            ref = oracle.predict(forest, model, frames[i], K, None if midp is None else midp[i], None if rot is None else rot[i])
            tag = f"case {case}: {w}x{h} patch {sw}x{sh} step {step} trees {trees} depth {depth} mixed {mixed} frame {i}"
            assert np.array_equal(leaf[i], ref.leaf_idx), tag
            assert np.array_equal(poses["mid_point"][i], ref.mid_point), (tag, poses["mid_point"][i], ref.mid_point)
            assert np.array_equal(poses["rotation"][i], ref.rotation), (tag, poses["rotation"][i], ref.rotation)


def test_state_across_batches(hp_mod):
    """One predictor, many batches of varying size and content (empty frames, half-empty frames): every pose
    equals the pose the same frame got in a reference pass -- no per-batch state (counters, tile flags, window
    lists, leaf histogram) leaks from one batch into the next."""
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 5, w=320, h=240, n_frames=24)
    model = synth.ModelParams(stepwidth=4)
    frames = synth.biwi_batch(24, 320, 240, first=40)
    frames[3] = 0
    frames[7, :, :160] = 0
    intr = hp_mod.IntrinsicMatrix(synth.default_intrinsic(320, 240))
    rs = np.random.RandomState(11)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        ref = hp.predict_batch(frames, intr).copy()
        for it in range(25):
            idx = rs.randint(0, 24, int(rs.randint(1, 25)))
            out = hp.predict_batch(frames[idx].copy(), intr)
            assert np.array_equal(out["mid_point"], ref["mid_point"][idx]), f"batch {it}"
            assert np.array_equal(out["rotation"], ref["rotation"][idx]), f"batch {it}"


def test_fewer_hits_than_vote_slices(hp_mod, oracle):
    """Frames with 1..6 hit records (one or two windows, three single-split trees whose leaves all
    vote): most of k_vote's 8 slices get no record but still own their share of the leaf histogram
    (rotation guess grid)."""
    nodes = np.zeros(3, dtype=NODE_DTYPE)
    nodes[0] = ((4, 4, 28, 28), (40, 40, 64, 64), 0.0, ~0, ~1)
    nodes[1] = ((10, 30, 34, 54), (44, 6, 68, 30), 10.0, ~2, ~3)
    nodes[2] = ((0, 0, 24, 24), (56, 56, 80, 80), -5.0, ~4, ~5)
    roots = np.array([0, 1, 2], dtype=np.int32)
    prob = np.array([1.0, 0.9, 0.95, 1.0, 0.8, 1.0])
    rs = np.random.RandomState(7)
    noff = [3, 4, 3, 5, 3, 4]
    begin = np.concatenate([[0], np.cumsum(noff)]).astype(np.uint32)
    offsets = (rs.uniform(-40, 40, (int(begin[-1]), 3)) + np.repeat(rs.uniform(-60, 60, (6, 3)), noff, axis=0)).astype(np.float32)
    rotations = rs.uniform(-3, 3, (int(begin[-1]), 3)) + np.repeat(rs.uniform(-40, 40, (6, 3)), noff, axis=0)
    forest = Forest(roots, nodes, prob, begin, begin.copy(), offsets, rotations)
    base = synth.biwi_like(640, 480, 777)
    ys, xs = np.nonzero(base)
    cy, cx = int(ys.mean()), int(xs.mean())
    seen = set()
    for w in (84, 88):                                                 # one window, two windows (the loops are exclusive, :546-548)
        frame = base[cy - 42:cy + 42, cx - 40:cx - 40 + w].copy()
        model = synth.ModelParams(stepwidth=4)
        K = synth.default_intrinsic(w, 84)
        with hp_mod.HoughPrediction(forest, model, device=0) as hp:
            hp.debug_enable(True)
            hp.predict_batch(frame[None].copy(), hp_mod.IntrinsicMatrix(K))
            seen.add(int(hp.debug_hit_counts(1)[0]))
        _check_frames(hp_mod, oracle, forest, model, frame[None].copy(), K, full=False)
    assert seen & set(range(1, 8)), f"hit counts {seen}: expected a frame with 1..7 records"


@pytest.mark.parametrize("w,h,step", [(1280, 720, 4), (1000, 600, 5)])
def test_frames_larger_than_vga(hp_mod, oracle, w, h, step):
    """Frame sizes beyond BASELINE's: more tiles per row than VGA, three-plus column parts in k_boxsum,
    a checkerboard of empty and occupied 640x480 blocks (tile flags next to each other)."""
    forest = synth.synth_forest(5, 9, synth.FOREST_SEED_BASE + 77)
    model = synth.ModelParams(stepwidth=step)
    base = synth.biwi_like(640, 480, 4711)
    frame = np.zeros((h, w), dtype=np.uint16)
    for oy in range(0, h, 480):
        for ox in range(0, w, 640):
            hh, ww = min(480, h - oy), min(640, w - ox)
            if (ox // 640 + oy // 480) % 2 == 0:
                frame[oy:oy + hh, ox:ox + ww] = base[:hh, :ww]
    _check_frames(hp_mod, oracle, forest, model, frame[None].copy(), synth.default_intrinsic(w, h), full=False)


@pytest.mark.parametrize("sw,sh,w,h,step", [(250, 16, 640, 64, 16), (16, 250, 80, 480, 9), (190, 190, 300, 280, 5), (8, 8, 64, 48, 1)])
def test_extreme_patch_shapes(hp_mod, oracle, sw, sh, w, h, step):
    """Very wide / very tall / near-maximal and tiny patches: tile geometry, segment counts and LDS
    carve-up at their edges.  (One window's summed-area table must fit the 160 KB LDS: patches beyond
    about 195x195 are refused with DH_ESIZE -- the reference's only trainer uses 80x80.)"""
    for mixed in (False, True):
        forest = synth.synth_forest(4, 7, 777 + sw + sh, patch=(sw, sh), rect_scale=0.25, rect_scale_max=0.7 if mixed else None)
        model = synth.ModelParams(stepwidth=step, subimage_width=sw, subimage_height=sh)
        frames = np.stack([synth.biwi_like(max(w, 96), max(h, 96), 4242 + i)[:h, :w] for i in range(2)]).copy()
        _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), full=False)


def test_patch_too_large_for_lds_is_refused(hp_mod):
    from depthhead_amd._lib import DepthheadError
    forest = synth.synth_forest(2, 3, 5, patch=(255, 255))
    model = synth.ModelParams(stepwidth=5, subimage_width=255, subimage_height=255)
    frames = np.zeros((1, 280, 300), dtype=np.uint16)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        with pytest.raises(DepthheadError) as ei:
            hp.predict_batch(frames, hp_mod.IntrinsicMatrix(synth.default_intrinsic(300, 280)))
        assert ei.value.code == -5 and "LDS" in str(ei.value)


def test_fitted_forest_config2(hp_mod, oracle):
    """The bench's own model (trainer-sized forest fitted to synthetic subjects: coherent votes,
    40 votes per leaf, mean shifts that move) on BASELINE config 2's geometry."""
    forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2, n_frames=16, subset=2500)
    model = synth.ModelParams(stepwidth=4)
    frames = synth.biwi_batch(4, 640, 480, first=200)
    K = synth.default_intrinsic()
    _check_frames(hp_mod, oracle, forest, model, frames, K)
    big = synth.biwi_batch(24, 640, 480, first=300)
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        poses = hp.predict_batch(big, hp_mod.IntrinsicMatrix(K))
    ref = oracle.predict_batch(forest, model, big, K)
    assert np.array_equal(poses["mid_point"], ref["mid_point"]) and np.array_equal(poses["rotation"], ref["rotation"])
    moved = sum(int(not np.array_equal(r, ref["rotation"][0])) for r in ref["rotation"])
    assert moved > 0                                          # not one constant answer


def test_error_behaviour_on_device(hp_mod, hip_lib):
    """Status codes instead of panics: taps before a batch, undersized tap buffers, NULL arguments,
    empty batches, a frame narrower than the patch, a rectangle that leaves the patch."""
    import ctypes as C
    from depthhead_amd import _lib
    from depthhead_amd._lib import DepthheadError
    forest = synth.synth_forest(3, 5, 31)
    model = synth.ModelParams(stepwidth=4)
    K = hp_mod.IntrinsicMatrix(synth.default_intrinsic(160, 120))
    with hp_mod.HoughPrediction(forest, model, device=0) as hp:
        with pytest.raises(DepthheadError) as ei:
            hp.debug_grids(1)                                   # no batch yet
        assert ei.value.code == -6
        assert hp.predict_batch(np.zeros((0, 120, 160), dtype=np.uint16), K).shape == (0,)   # empty batch is fine
        with pytest.raises(DepthheadError) as ei:
            hp.predict_batch(np.zeros((1, 60, 160), dtype=np.uint16), K)        # frame shorter than the 80-px patch
        assert ei.value.code == -5 and "smaller than" in str(ei.value)
        frames = synth.biwi_batch(2, 160, 120)
        hp.predict_batch(frames, K)
        with pytest.raises(DepthheadError) as ei:
            hp.debug_leaf_indices(2, 160, 120)                  # taps were not enabled for that batch
        assert ei.value.code == -6
        hp.debug_enable(True)
        hp.predict_batch(frames, K)
        small = np.zeros(3, dtype=np.int32)
        assert hip_lib.dh_debug_leaf_indices(hp._ph, small.ctypes.data_as(C.c_void_p), C.c_size_t(3)) == -1
        assert hip_lib.dh_predict_batch(hp._ph, None, 1, 160, 120, None, None, None, None, None) == -1
        assert hip_lib.dh_graph_launch(hp._ph, None) == -6      # nothing captured
        cnt = C.c_size_t()
        assert hip_lib.dh_debug_votes(hp._ph, 5, 0, None, C.c_size_t(0), C.byref(cnt)) == -1   # frame out of range
    wide = synth.synth_forest(2, 3, 32, patch=(100, 100))       # rectangles reach x = 100
    with pytest.raises(DepthheadError) as ei:
        hp_mod.HoughPrediction(wide, synth.ModelParams())       # 80x80 patch
    assert ei.value.code == -2 and "leaves the" in str(ei.value)


def test_cpp_example_runs(hp_mod, tmp_path):
    """The pure C++ host of examples/predict_frame.cpp (no Python, no PyTorch in the process) builds, runs on the
    GPU and prints a pose for its synthetic head at 900 mm."""
    import os, re, shutil, subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "depthhead_amd")
    exe = str(tmp_path / "predict_frame")
    subprocess.run([gxx, "-std=c++17", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "predict_frame.cpp"),
                    "-L" + libdir, "-ldepthhead_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    m = re.search(r"mid_point = \((-?\d+), (-?\d+), (-?\d+)\) mm", res.stdout)
    assert m, res.stdout
    assert 700 <= int(m.group(3)) <= 1100, res.stdout          # the blob's surface is at 805..900 mm; votes point 20..24 mm behind it


@pytest.mark.parametrize("w,h,sw,sh,step", [(27, 50, 19, 49, 6), (40, 33, 24, 24, 3), (90, 30, 30, 17, 5)])
def test_tiny_frames_general_path(hp_mod, oracle, w, h, sw, sh, step):
    """Footprints whose summed-area table is smaller than the scratch of the register-scan build (found by
    tools/fuzz_parity.py case 131819): the pass-based build must take over."""
    forest = synth.synth_forest(8, 2, synth.FOREST_SEED_BASE + 400 + w, patch=(sw, sh), rect_scale=0.3, rect_scale_max=0.8)
    model = synth.ModelParams(stepwidth=step, subimage_width=sw, subimage_height=sh)
    frames = np.stack([synth.biwi_like(96, 96, 5200 + i)[20:20 + h, 30:30 + w] for i in range(3)]).copy()
    _check_frames(hp_mod, oracle, forest, model, frames, synth.default_intrinsic(w, h), full=False)
