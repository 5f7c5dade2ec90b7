"""Host-side logic that needs no GPU: generators, sharding, vote aggregation."""
import numpy as np
import pytest

from depthhead_amd import synth
from depthhead_amd.dist import shard_range
from depthhead_amd.forest import Forest
from depthhead_amd.prediction import aggregate_votes


def test_generators_are_pure_functions_of_the_seed():
    a, b = synth.biwi_like(160, 120, 123), synth.biwi_like(160, 120, 123)
    assert np.array_equal(a, b) and not np.array_equal(a, synth.biwi_like(160, 120, 124))
    f1, f2 = synth.synth_forest(3, 6, 11), synth.synth_forest(3, 6, 11)
    assert f1.nodes.tobytes() == f2.nodes.tobytes() and np.array_equal(f1.offsets, f2.offsets)
    frac = (synth.biwi_like(640, 480, synth.FRAME_SEED_BASE) > 0).mean()
    assert 0.05 < frac < 0.45      # BIWI-like: subject at ~1 m in front of a thresholded background


def test_forest_structure_invariants():
    f = synth.synth_forest(4, 9, 5)
    assert f.max_depth() == 9 and f.n_trees == 4
    cz, co = f.nodes["child_zero"], f.nodes["child_one"]
    kids = np.concatenate([cz, co])
    assert sorted(kids[kids >= 0].tolist() + f.roots.tolist()) == list(range(f.n_nodes))   # every node has one parent
    assert sorted((~kids[kids < 0]).tolist()) == list(range(f.n_leaves))                   # every leaf has one parent
    assert np.all(f.nodes["r1"][:, 2] <= 80) and np.all(f.nodes["r2"][:, 3] <= 80)
    pos = f.leaf_prob > 0
    n = np.diff(f.off_begin.astype(np.int64))
    assert np.all(n[pos] >= 2) and np.all(n[~pos] == 0)


def test_forest_save_load_roundtrip(tmp_path):
    f = synth.synth_forest(2, 5, 9)
    p = str(tmp_path / "forest.npz")
    f.save(p)
    g = Forest.load(p)
    assert g.nodes.tobytes() == f.nodes.tobytes() and np.array_equal(g.rotations, f.rotations) and np.array_equal(g.roots, f.roots)


@pytest.mark.parametrize("n,world", [(4096, 8), (256, 1), (10, 3), (3, 8), (0, 4), (1000, 7)])
def test_shard_range_partitions_frames(n, world):
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 3, 3)


def test_aggregate_votes_wraps_like_u32():
    v = np.array([[1, 2, 3, 5], [1, 2, 3, 7], [-4, 0, 9, 1], [1, 2, 3, -1], [0, 0, 0, 2**31 - 1], [0, 0, 0, 2**31 - 1], [0, 0, 0, 2]],
                 dtype=np.int32)
    out = aggregate_votes(v)
    assert out.tolist() == [[-4, 0, 9, 1], [0, 0, 0, 0], [1, 2, 3, 11]]    # 5+7+0xFFFFFFFF wraps to 11; 2*(2^31-1)+2 wraps to 0
    assert aggregate_votes(np.zeros((0, 4), dtype=np.int32)).shape == (0, 4)
