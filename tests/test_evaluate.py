"""Evaluation aggregation (examples/db_evaluate.rs:219-313 + the error definitions of eval_files/overview.html): the
host bookkeeping on CPU with a stand-in predictor, the whole loop on a synthetic BIWI directory tree on the GPU."""
import json
import os
import struct

import numpy as np
import pytest

from depthhead_amd import biwi, evaluate, synth
from depthhead_amd._lib import POSE_DTYPE


class _Fixed:
    """Predictor stand-in: returns prescribed poses (host bookkeeping tests need no GPU)."""

    def __init__(self, mids, rots):
        self.mids, self.rots, self.calls = list(mids), list(rots), []

    def predict_batch(self, frames, intr):
        n = len(frames)
        self.calls.append(n)
        out = np.zeros(n, dtype=POSE_DTYPE)
        for i in range(n):
            out["mid_point"][i] = self.mids.pop(0)
            out["rotation"][i] = self.rots.pop(0)
        return out


def _truth(i, shape=(8, 8), K=None):
    K = np.eye(3, dtype=np.float32) if K is None else K
    gt = biwi.GT(np.array([10.0 * i, 0.0, 1000.0], np.float32), np.zeros(2, np.float32), np.array([float(i), 0.0, 0.0], np.float32))
    return biwi.DepthTrue(gt, np.zeros(shape, np.uint16), "", K, f"frame_{i:05d}")


def test_entry_bookkeeping_and_metrics():
    frames = [_truth(i) for i in range(5)] + [_truth(5, (9, 8))] + [_truth(6, (9, 8), np.eye(3, dtype=np.float32) * 2)]
    mids = [[10.0 * i + 3.0, 4.0, 1000.0] for i in range(7)]                # every guess is 5 mm off (3-4-5)
    rots = [[(i + 2.0) * evaluate.PI_REF / 180.0, 0.0, 0.0] for i in range(7)]   # 2 degrees off
    fx = _Fixed(mids, rots)
    e = evaluate.EvalEntry()
    e.eval(iter(frames), fx, batch=3)
    assert fx.calls == [3, 2, 1, 1]                                         # batches break at the size, at a new shape, at a new intrinsic
    res = evaluate.EvaluationResult([1], "forest.json", [(1, e)])
    assert np.allclose(res.distances("midpoint"), 5.0) and np.allclose(res.distances("rot"), 2.0, atol=1e-5)
    assert np.isclose(res.mean_square_error("midpoint"), 25.0 / 3) and np.isclose(res.mean_square_error("rot"), 4.0 / 3, atol=1e-5)
    th, acc = res.accuracy(res.distances("midpoint"))
    assert list(th[:3]) == [0, 5, 10] and acc[0] == 0.0 and acc[1] == 1.0    # `<=` threshold (overview.html:85)
    assert np.allclose(res.coord_distances("midpoint", 0), 3.0)
    doc = json.loads(res.to_json())                                          # the reference's JSON shape
    assert doc["persons"] == [1] and doc["res"][0][0] == 1 and set(doc["res"][0][1]) == {"guess_midpoint", "guess_rot", "truth_midpoint", "truth_rot"}
    assert doc["res"][0][1]["guess_rot"][3][0] == pytest.approx(5.0, abs=1e-5)
    assert res.summary()["frames"] == 7


def test_r2d_uses_the_references_pi():
    assert evaluate.r2d(3.14159)[()] == np.float32(180.0) and evaluate.r2d([0.5]).dtype == np.float32


@pytest.mark.gpu
def test_evaluate_synthetic_biwi_tree(tmp_path, hip_lib, oracle):
    """A BIWI-shaped directory tree written with the package's own encoders, two subjects, evaluated end to end
    (directory walk -> RLE decode -> GPU batches -> aggregation) in both the 3-D and the 2-D Hough variant."""
    from depthhead_amd import prediction
    w, h = 320, 240
    K = synth.default_intrinsic(w, h)
    cal = "\n".join(" ".join(f"{v:g}" for v in row) for row in K) + "\n"
    for d in ("head_pose_masks", "hpdb", "db_annotations"):
        for p in (1, 2):
            os.makedirs(tmp_path / d / f"{p:02d}")
    truth = {}
    for p in (1, 2):
        (tmp_path / "hpdb" / f"{p:02d}" / "depth.cal").write_text(cal)
        for i in range(5):
            img = synth.biwi_like(w, h, synth.FRAME_SEED_BASE + 700 + 10 * p + i)
            name = f"frame_{i:05d}"
            (tmp_path / "hpdb" / f"{p:02d}" / f"{name}_depth.bin").write_bytes(biwi.encode_depth(img))
            if i == 3:
                continue                                                     # no mask: the reference skips the frame (biwi.rs:281-284)
            (tmp_path / "head_pose_masks" / f"{p:02d}" / f"{name}_depth_mask.png").write_bytes(b"")
            gt = np.array([5.0 * i, -3.0, 900.0 + p, 1.0, 2.0, 3.0], dtype="<f4")
            (tmp_path / "db_annotations" / f"{p:02d}" / f"{name}_pose.bin").write_bytes(gt.tobytes())
            truth[(p, i)] = (img, gt)
    reader = biwi.BiwiReader(str(tmp_path / "head_pose_masks"), str(tmp_path / "hpdb"), str(tmp_path / "db_annotations"))
    assert reader.is_valid() and reader.person_count() == 2
    forest = synth.fit_forest(6, 10, synth.FOREST_SEED_BASE + 170, n_frames=12, subset=1500)
    model = synth.ModelParams(stepwidth=4)
    with prediction.HoughPrediction(forest, model, device=0) as hp:
        res = evaluate.EvaluationResult([2, 1], "synthetic")
        res.evaluate(reader, hp, batch=3)
        res2d = evaluate.EvaluationResult([1], "synthetic")
        res2d.evaluate(reader, hp, from2d=True)
    assert [p for p, _ in res.res] == [2, 1] and all(len(e.guess_midpoint) == 4 for _, e in res.res)
    for p, e in res.res:
        for j, i in enumerate((0, 1, 2, 4)):
            img, gt = truth[(p, i)]
            ref = oracle.predict(forest, model, img, K, taps=False)
            assert e.guess_midpoint[j] == [float(v) for v in ref.mid_point]
            assert e.guess_rot[j] == [float(v) for v in evaluate.r2d(ref.rotation)]
            assert e.truth_midpoint[j] == [float(v) for v in gt[:3]] and e.truth_rot[j] == [float(v) for v in gt[3:]]
    for j, i in enumerate((0, 1, 2, 4)):
        img, _ = truth[(1, i)]
        mid, _rot = oracle.predict_from2dhough(forest, model, img, K)
        assert res2d.res[0][1].guess_midpoint[j] == [float(v) for v in mid] and res2d.res[0][1].guess_rot[j] == [0.0, 0.0, 0.0]
    s = res.summary()
    assert s["frames"] == 8 and s["midp_mse"] >= 0


def test_evaluation_json_maps_non_finite_values_to_null():
    """serde_json writes `null` for a non-finite f32 / f64; a bare NaN / Infinity (Python's default) would make the reference's
    report page reject the whole file."""
    import json
    from depthhead_amd.evaluate import EvalEntry, EvaluationResult
    e = EvalEntry()
    e.guess_midpoint = [[1.0, float("nan"), 3.0], [float("inf"), -2.0, float("-inf")]]
    e.guess_rot = [[0.0, 1.5, -1.5], [0.0, 0.0, 0.0]]
    e.truth_midpoint = [[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]
    e.truth_rot = [[0.0, 1.0, -1.0], [0.0, 0.0, 0.0]]
    r = EvaluationResult(persons=[7], trained_tree_path="forest.json", res=[(7, e)])
    text = r.to_json()
    assert "NaN" not in text and "Infinity" not in text
    doc = json.loads(text)
    assert doc["persons"] == [7] and doc["res"][0][0] == 7
    assert doc["res"][0][1]["guess_midpoint"] == [[1.0, None, 3.0], [None, -2.0, None]]
    assert doc["res"][0][1]["truth_rot"] == [[0.0, 1.0, -1.0], [0.0, 0.0, 0.0]]
