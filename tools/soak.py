"""Soak: many batches of varying size / offset through one predictor; every pose must equal the pose the
same frame got in a reference pass (catches stale per-batch state: counters, tile flags, window lists)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
model = synth.ModelParams(stepwidth=4)
frames = synth.biwi_batch(96)
frames[5] = 0
frames[17, :, :320] = 0
intr = IntrinsicMatrix(synth.default_intrinsic(640, 480))
rs = np.random.RandomState(3)
with HoughPrediction(forest, model, device=0) as hp:
    ref = hp.predict_batch(frames, intr).copy()
    bad = 0
    for it in range(150):
        n = int(rs.randint(1, 97))
        idx = rs.randint(0, 96, n)
        out = hp.predict_batch(frames[idx].copy(), intr)
        if not (np.array_equal(out["mid_point"], ref["mid_point"][idx]) and np.array_equal(out["rotation"], ref["rotation"][idx])):
            bad += 1
            print("MISMATCH at iteration", it, "n", n)
    print("soak done, mismatching batches:", bad)
