"""Ad-hoc parity check on a 12-megapixel frame (index arithmetic beyond 2^24 pixels), GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix
from oracle import pyoracle

for (w, h, step, mixed) in ((4096, 3072, 16, False), (4096, 3072, 12, True), (8192, 600, 8, False)):
    forest = synth.synth_forest(4, 8, synth.FOREST_SEED_BASE + 91, rect_scale=0.3, rect_scale_max=0.7 if mixed else None)
    model = synth.ModelParams(stepwidth=step)
    base = synth.biwi_like(640, 480, 4711)
    frame = np.zeros((h, w), dtype=np.uint16)
    for oy in range(0, h, 480):
        for ox in range(0, w, 640):
            hh, ww = min(480, h - oy), min(640, w - ox)
            if (ox // 640 + oy // 480) % 3 == 0:
                frame[oy:oy + hh, ox:ox + ww] = base[:hh, :ww]
    K = synth.default_intrinsic(w, h)
    with HoughPrediction(forest, model, device=0) as hp:
        hp.debug_enable(True)
        poses = hp.predict_batch(np.stack([frame, frame]), IntrinsicMatrix(K))
        leaf = hp.debug_leaf_indices(2, w, h)
    ref = pyoracle.predict(forest, model, frame, K, cell_cap=1 << 25)
    ok = all(np.array_equal(leaf[i], ref.leaf_idx) and np.array_equal(poses["mid_point"][i], ref.mid_point) and
             np.array_equal(poses["rotation"][i], ref.rotation) for i in range(2))
    print(w, h, step, "mixed" if mixed else "uniform", "OK" if ok else "MISMATCH", flush=True)
