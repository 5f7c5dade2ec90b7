"""Evaluation aggregation over BIWI subjects: the host loop of
`/root/reference/examples/db_evaluate.rs` (`EvalEntrie::eval` :279-313, `EvaluationResult` :219-258) with the
per-frame `predict_parameter` call replaced by frame BATCHES through the GPU predictor, and the error definitions the
reference's report page computes from `evaluation.json` (`examples/eval_files/overview.html`: `compDistance` :175-186,
`compAccuracy` :82-90, `meansqareerror` :236-251, thresholds 0, 5, ..., 95 :254-257).

The reference evaluates with the serial `predict_parameter` (db_evaluate.rs:296); batches here go through the
`predict_parameter_parallel`-equivalent GPU path, which is numerically identical by construction (prediction.rs:386 vs :407:
only the forest call differs, and the parity tests hold both to the same oracle).

The JSON this writes has the reference's shape -- `{"persons": [...], "trained_tree_path": "...", "res": [[person,
{"guess_midpoint": [[f32; 3]...], "guess_rot": ..., "truth_midpoint": ..., "truth_rot": ...}], ...]}` (serde writes the
`(usize, EvalEntrie)` tuples as 2-element arrays) -- so the reference's own `overview.html` renders it.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np

from .prediction import HoughPrediction, IntrinsicMatrix

PI_REF = 3.14159   # the reference's literal (db_evaluate.rs:284-286)


def r2d(x) -> np.ndarray:
    """`r2d!`: `($x * 180.0 / 3.14159) as f32` on the f64 rotation (db_evaluate.rs:284-286)."""
    return (np.asarray(x, dtype=np.float64) * 180.0 / PI_REF).astype(np.float32)


@dataclass
class EvalEntry:
    """db_evaluate.rs:262-267 `EvalEntrie`: one subject."""
    guess_midpoint: list = field(default_factory=list)
    guess_rot: list = field(default_factory=list)
    truth_midpoint: list = field(default_factory=list)
    truth_rot: list = field(default_factory=list)

    def eval(self, frames_iter, hp: HoughPrediction, from2dhough: bool = False, batch: int = 64) -> None:
        """db_evaluate.rs:279-313: every frame of `frames_iter` (objects with `.depth`, `.intrinsic`, `.trans.pos3d`,
        `.trans.rot`, e.g. `biwi.DepthTrue`) is predicted with no guesses; consecutive frames that share size and intrinsic
        go to the GPU as one batch."""
        pend: list = []

        def flush():
            if not pend:
                return
            frames = np.stack([t.depth for t in pend])
            intr = IntrinsicMatrix(pend[0].intrinsic)
            poses = hp.predict_parameter_from2dhough(frames, intr) if from2dhough else hp.predict_batch(frames, intr)
            for t, p in zip(pend, poses):
                self.guess_midpoint.append([float(v) for v in p["mid_point"]])
                self.guess_rot.append([float(v) for v in r2d(p["rotation"])])
                self.truth_midpoint.append([float(v) for v in np.asarray(t.trans.pos3d, dtype=np.float32)])
                self.truth_rot.append([float(v) for v in np.asarray(t.trans.rot, dtype=np.float32)])
            pend.clear()

        for t in frames_iter:
            if pend and (t.depth.shape != pend[0].depth.shape or not np.array_equal(t.intrinsic, pend[0].intrinsic) or len(pend) >= batch):
                flush()
            pend.append(t)
        flush()

    def as_dict(self) -> dict:
        return {"guess_midpoint": self.guess_midpoint, "guess_rot": self.guess_rot,
                "truth_midpoint": self.truth_midpoint, "truth_rot": self.truth_rot}


@dataclass
class EvaluationResult:
    """db_evaluate.rs:219-258."""
    persons: list
    trained_tree_path: str
    res: list = field(default_factory=list)     # [(person, EvalEntry)]

    def evaluate(self, reader, hp: HoughPrediction, from2d: bool = False, batch: int = 64) -> None:
        for person in self.persons:
            entry = EvalEntry()
            entry.eval(reader.person(person), hp, from2d, batch)
            self.res.append((person, entry))

    def to_json(self) -> str:
        """The file `eval_files/overview.html` loads.  serde_json writes `null` for a non-finite f32 / f64 (a pose component
        can be inf when the frame's depth at the 2-D argmax is 0); Python's encoder would write a bare NaN / Infinity, which
        `JSON.parse` rejects -- so non-finite values become null here too, and `allow_nan=False` makes a miss loud."""
        def clean(x):
            if isinstance(x, (list, tuple)):
                return [clean(v) for v in x]
            if isinstance(x, dict):
                return {k: clean(v) for k, v in x.items()}
            if isinstance(x, (float, np.floating)):
                return float(x) if np.isfinite(x) else None
            if isinstance(x, np.integer):
                return int(x)
            return x
        return json.dumps(clean({"persons": list(self.persons), "trained_tree_path": self.trained_tree_path,
                                 "res": [[p, e.as_dict()] for p, e in self.res]}), allow_nan=False)

    # ---- the report page's error definitions (overview.html) -------------------------------------------
    def _all(self, source: str):
        g = [np.asarray(getattr(e, "guess_" + source), dtype=np.float64).reshape(-1, 3) for _, e in self.res]
        t = [np.asarray(getattr(e, "truth_" + source), dtype=np.float64).reshape(-1, 3) for _, e in self.res]
        return (np.concatenate(g) if g else np.zeros((0, 3))), (np.concatenate(t) if t else np.zeros((0, 3)))

    def distances(self, source: str = "midpoint") -> np.ndarray:
        """`allDistances` / `compDistance`: Euclidean distance per frame (mm for midpoint, degrees for rot)."""
        g, t = self._all(source)
        return np.sqrt(((g - t) ** 2).sum(axis=1))

    def coord_distances(self, source: str = "midpoint", idx: int = 0) -> np.ndarray:
        """`coordDistance`: |guess - truth| of one coordinate."""
        g, t = self._all(source)
        return np.abs(g[:, idx] - t[:, idx])

    def mean_square_error(self, source: str = "midpoint") -> float:
        """`meansqareerror`: sum of squared coordinate errors / 3 / number of frames."""
        g, t = self._all(source)
        return float(((t - g) ** 2).sum() / 3 / len(t)) if len(t) else float("nan")

    @staticmethod
    def accuracy(distances, thresholds=None):
        """`plotAccuracy` / `compAccuracy`: share of frames with distance <= threshold, thresholds 0, 5, ..., 95."""
        th = np.arange(20) * 5 if thresholds is None else np.asarray(thresholds)
        d = np.asarray(distances, dtype=np.float64)
        return th, np.array([(d <= x).mean() if d.size else np.nan for x in th])

    def summary(self) -> dict:
        dm, dr = self.distances("midpoint"), self.distances("rot")
        return {"frames": int(dm.size), "midp_mse": self.mean_square_error("midpoint"), "rot_mse": self.mean_square_error("rot"),
                "midpoint_mean_distance_mm": float(dm.mean()) if dm.size else float("nan"),
                "rot_mean_distance_deg": float(dr.mean()) if dr.size else float("nan"),
                "midpoint_accuracy": dict(zip(*[x.tolist() for x in self.accuracy(dm)])),
                "rot_accuracy": dict(zip(*[x.tolist() for x in self.accuracy(dr)]))}
