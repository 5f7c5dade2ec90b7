"""The oracle against the committed fixtures, in both rectangle modes (the faithful O(area) loops
of src/types.rs:317-339 and the summed-area-table shortcut must agree bit for bit)."""
import numpy as np
import pytest

import golden_util


@pytest.mark.parametrize("name", golden_util.names())
def test_oracle_reproduces_golden(oracle, name):
    forest, model, frames, K, exp = golden_util.load(name)
    assert len(exp) == frames.shape[0] and len(exp) > 0
    for mode in (oracle.RECT_FAITHFUL, oracle.RECT_SAT):
        for i, e in enumerate(exp):
            r = oracle.predict(forest, model, frames[i], K, rect_mode=mode)
            assert np.array_equal(r.leaf_idx, e["leaf_idx"])
            assert np.array_equal(r.patch_flags, e["patch_flags"])
            assert np.array_equal(r.pos_grid, e["pos_grid"]) and np.array_equal(r.rot_grid, e["rot_grid"])
            assert np.array_equal(np.concatenate([r.guess_mid, r.guess_rot]), e["guess"])
            assert np.array_equal(r.mid_cells, e["mid_cells"]) and np.array_equal(r.rot_cells, e["rot_cells"])
            assert np.array_equal(r.ms_trace_mid, e["ms_trace_mid"]) and np.array_equal(r.ms_trace_rot, e["ms_trace_rot"])
            assert np.array_equal(r.mid_point, e["mid_point"]) and np.array_equal(r.rotation, e["rotation"])


def test_golden_cases_are_not_trivial():
    """Every fixture must exercise the whole path: gated patches, votes in both accumulators and a
    mean shift that actually moves."""
    moved = 0
    for name in golden_util.names():
        _, _, _, _, exp = golden_util.load(name)
        for e in exp:
            assert (e["patch_flags"] & 2).any(), name
            assert len(e["mid_cells"]) > 0 and len(e["rot_cells"]) > 0, name
            moved += int(len(e["ms_trace_rot"]) > 1 and not np.array_equal(e["ms_trace_rot"][0], e["ms_trace_rot"][-1]))
    assert moved > 0


def test_batch_api_matches_single(oracle):
    forest, model, frames, K, exp = golden_util.load(golden_util.names()[0])
    poses = oracle.predict_batch(forest, model, frames, K, threads=2)
    for i, e in enumerate(exp):
        assert np.array_equal(poses["mid_point"][i], e["mid_point"]) and np.array_equal(poses["rotation"][i], e["rotation"])
