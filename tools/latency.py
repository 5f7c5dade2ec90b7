"""Single-frame latency through the host entry point (what the Rust shim of INTEGRATION.md calls per frame)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
for stride in (4, 10):
    model = synth.ModelParams(stepwidth=stride)
    frames = synth.biwi_batch(8)
    intr = IntrinsicMatrix(synth.default_intrinsic(640, 480))
    with HoughPrediction(forest, model, device=0) as hp:
        for i in range(20):
            hp.predict_batch(frames[i % 8][None], intr)
        t0 = time.perf_counter()
        N = 300
        for i in range(N):
            hp.predict_batch(frames[i % 8][None], intr)
        dt = (time.perf_counter() - t0) / N
        print(f"stride {stride}: host-buffer single frame {dt * 1e3:.3f} ms per call ({1 / dt:.0f} frames/s), pageable input")
