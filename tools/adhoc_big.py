"""Ad-hoc parity check on frame sizes beyond the test-suite's (GPU box): HIP path vs oracle (SAT mode)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix
from oracle import pyoracle

for (w, h, step, scale) in ((1280, 720, 4, 0.3), (1920, 1080, 8, 0.3), (1000, 600, 5, 0.25)):
    forest = synth.synth_forest(5, 9, synth.FOREST_SEED_BASE + 77, rect_scale=scale)
    model = synth.ModelParams(stepwidth=step)
    base = synth.biwi_like(640, 480, 4711)
    frame = np.zeros((h, w), dtype=np.uint16)
    for oy in range(0, h, 480):
        for ox in range(0, w, 640):
            hh, ww = min(480, h - oy), min(640, w - ox)
            if (ox // 640 + oy // 480) % 2 == 0:
                frame[oy:oy + hh, ox:ox + ww] = base[:hh, :ww]
    K = synth.default_intrinsic(w, h)
    with HoughPrediction(forest, model, device=0) as hp:
        hp.debug_enable(True)
        poses = hp.predict_batch(frame[None], IntrinsicMatrix(K))
        leaf = hp.debug_leaf_indices(1, w, h)
    ref = pyoracle.predict(forest, model, frame, K)
    ok = np.array_equal(leaf[0], ref.leaf_idx) and np.array_equal(poses["mid_point"][0], ref.mid_point) and np.array_equal(poses["rotation"][0], ref.rotation)
    print(w, h, step, "OK" if ok else "MISMATCH", poses["mid_point"][0], ref.mid_point)
