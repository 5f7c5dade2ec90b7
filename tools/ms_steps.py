"""Diagnostic: how many mean-shift position updates the bench frames need (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
model = synth.ModelParams(stepwidth=4)
frames = synth.biwi_batch(64)
with HoughPrediction(forest, model, device=0) as hp:
    hp.debug_enable(True)
    hp.predict_batch(frames, IntrinsicMatrix(synth.default_intrinsic(640, 480)))
    hc = hp.debug_hit_counts(64)
    for which, name in ((0, "mid"), (1, "rot")):
        tr, st = hp.debug_meanshift(64, which)
        print(name, "steps: mean %.1f min %d max %d" % (st.mean(), st.min(), st.max()), np.bincount(st))
    print("hits per frame: mean %.0f min %d max %d" % (hc.mean(), hc.min(), hc.max()))
    tr, st = hp.debug_meanshift(64, 1)
    for i in range(4):
        print("rot trace frame", i, tr[i].tolist())
    tr, st = hp.debug_meanshift(64, 0)
    for i in range(2):
        print("mid trace frame", i, tr[i].tolist())
