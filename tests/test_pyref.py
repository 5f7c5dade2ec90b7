"""The second, independent restatement (oracle/pyref.py: plain Python / numpy scalars, written from the Rust text) against the C
oracle and the committed goldens, every intermediate: leaf indices, patch flags, both guess grids, the guesses, both sparse
accumulators, both mean-shift traces, the pose.  It pins nothing to the reference (PARITY UNPINNED: the reference cannot be
run here and holds no fixture for these stages); it is the defence against a misreading shared by one restatement and the
kernels that were tested against it."""
import numpy as np
import pytest

import golden_util
from oracle import pyref
from test_hand_case import EXPECT_MID, EXPECT_ROT, hand_case

KEYS = ("leaf_idx", "patch_flags", "pos_grid", "rot_grid", "mid_cells", "rot_cells", "ms_trace_mid", "ms_trace_rot", "mid_point", "rotation")


def _same(a: dict, r):
    for k in KEYS:
        assert np.array_equal(a[k], getattr(r, k)), k
    assert np.array_equal(a["guess_mid"], r.guess_mid) and np.array_equal(a["guess_rot"], r.guess_rot)


@pytest.mark.parametrize("name", golden_util.names())
def test_pyref_reproduces_the_goldens_and_the_c_oracle(oracle, name):
    forest, model, frames, K, exp = golden_util.load(name)
    for i, e in enumerate(exp):
        a = pyref.predict(forest, model, frames[i], K)
        for k in KEYS:
            assert np.array_equal(a[k], e[k]), (name, i, k)
        assert np.array_equal(np.concatenate([a["guess_mid"], a["guess_rot"]]), e["guess"])
        _same(a, oracle.predict(forest, model, frames[i], K, rect_mode=oracle.RECT_FAITHFUL))


def test_pyref_on_the_paper_case(oracle):
    forest, model, img, K = hand_case()
    a = pyref.predict(forest, model, img, K)
    assert a["leaf_idx"].tolist() == [[0]] and a["patch_flags"].tolist() == [3]
    assert a["rot_cells"].tolist() == [[63, 66, 70, 500], [64, 67, 70, 500]]
    assert a["mid_cells"].tolist() == [[-11, 21, 398, 500], [-10, 20, 399, 500]]
    assert a["guess_rot"].tolist() == [63, 69, 69] and a["guess_mid"].tolist() == [-25, 46, 1000]
    assert np.array_equal(a["mid_point"], EXPECT_MID) and np.array_equal(a["rotation"], EXPECT_ROT)
    _same(a, oracle.predict(forest, model, img, K))


def test_pyref_with_guess_overrides(oracle):
    """The Option<> arguments of predict_parameter_generic (prediction.rs:437-460): casts, the pi literal, NaN and huge guesses."""
    forest, model, frames, K, _ = golden_util.load(golden_util.names()[0])
    for mg, rg in (([12.7, -300.2, 905.9], [0.3, -0.2, 1.1]), ([float("nan"), 1e12, -1e12], None), (None, [float("nan"), 50.0, -50.0])):
        a = pyref.predict(forest, model, frames[0], K, mg, rg)
        _same(a, oracle.predict(forest, model, frames[0], K, None if mg is None else np.array(mg, dtype=np.float32), None if rg is None else np.array(rg)))


def test_pyref_helpers_against_the_reference_kats():
    """The reference's OWN known answers for the helpers -- test_intrinsic (src/types.rs:476-488), test_mean_cov3 / test_det_2_3_trace /
    test_inverse / test_mat_vec_mul (src/meancov_estimation.rs:461-525), same numbers and tolerances -- through this restatement
    as well (tests/test_oracle_kat.py holds the C oracle to them)."""
    F32, F64 = np.float32, np.float64
    # types.rs:476-488
    intr = pyref.Intrinsic([[22.0, 11.4, 12.11], [2.1, 4.1, 2.11], [1.3, 3.1, 19.0]])
    p2 = intr.space_to_img([F32(11.0), F32(12.0), F32(32.2)])
    assert abs(float(p2[0]) - 1.15896578) < 1e-4 and abs(float(p2[1]) - 0.21143073) < 1e-4
    back = intr.img_to_space(p2, F32(32.2))
    assert all(abs(float(b) - e) < 1e-4 for b, e in zip(back, (11.0, 12.0, 32.2)))
    # meancov_estimation.rs:461-490 (the trace is what the path uses: sum of the three diagonal answers)
    v = np.array([[1.0, 2.0, 3.0], [1.2, 1.0, 3.2], [-1.0, -2.1, 3.0], [0.0, 1.0, 0.0]])
    assert abs(float(pyref.trace_of_cov(v, F64)) - (1.0266666 + 3.1691666 + 2.36)) < 3e-3
    v = np.array([[-32.48225021362305, 24.72743034362793, -3.9425208568573], [-25.82341957092285, -25.307233810424805, 1.955498456954956],
                  [35.37421417236328, -18.529083251953125, -5.888242721557617], [43.30265808105469, -60.69481658935547, -15.176074028015137],
                  [32.97354507446289, -7.171285629272461, -3.897606134414673]])
    assert abs(float(pyref.trace_of_cov(v, F64)) - (1341.63076476 + 954.396794746 + 38.573568346)) < 3e-3
    assert np.isnan(pyref.trace_of_cov(np.array([[1.0, 2.0, 3.0]]), F64))            # n = 1: 0 / 0 (a one-vote leaf never votes)
    # :492-500, :502-515, :517-525
    assert abs(float(pyref.mat3_det([[1.0, 3.0, 22.0], [2.0, 44.0, 1.0], [2.0, 0.0, 3.1]], F64)) - -1812.199) < 1e-3
    inv = pyref.mat3_inv([[2.3, 1.4, 12.11], [2.1, 44.11, 2.11], [1.3, 4.1, 19.0]], F64)
    want = [[0.65540671, 0.01821446, -0.4197583], [-0.02936075, 0.02209108, 0.01626034], [-0.03850788, -0.00601328, 0.07784307]]
    assert all(abs(float(inv[i][j]) - want[i][j]) < 1e-3 for i in range(3) for j in range(3))
    mv = pyref.mat_vec([[1.3, 12.1, 2.3], [3.1, 33.1, 14.1], [1.0, 2.0, 3.0]], [11.0, 12.0, 32.2], F64)
    assert all(abs(float(a) - b) < 1e-3 for a, b in zip(mv, (233.56, 885.32, 131.6)))
