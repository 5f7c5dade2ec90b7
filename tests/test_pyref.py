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
    """The reference's own known answers for the helpers (src/types.rs:476-488, src/meancov_estimation.rs:450-533), through
    this restatement as well."""
    F32, F64 = np.float32, np.float64
    intr = pyref.Intrinsic([[560.0, 0.0, 320.0], [0.0, 560.0, 240.0], [0.0, 0.0, 1.0]])
    p = intr.img_to_space([F32(100.0), F32(50.0)], F32(800.0))
    back = intr.space_to_img(p)
    assert abs(float(back[0]) - 100.0) < 1e-3 and abs(float(back[1]) - 50.0) < 1e-3 and float(p[2]) == 800.0
    m = [[1.0, 2.0, 3.0], [0.0, 1.0, 4.0], [5.0, 6.0, 0.0]]
    assert float(pyref.mat3_det(m, F64)) == 1.0
    inv = pyref.mat3_inv(m, F64)
    assert [[float(x) for x in r] for r in inv] == [[-24.0, 18.0, 5.0], [20.0, -15.0, -4.0], [-5.0, 4.0, 1.0]]
    assert float(pyref.trace_of_cov(np.array([[10.0, 20.0, 30.0], [12.0, 22.0, 32.0]]), F64)) == 6.0
    assert np.isnan(pyref.trace_of_cov(np.array([[1.0, 2.0, 3.0]]), F64))            # n = 1: 0 / 0
