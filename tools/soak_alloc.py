"""Soak, fourth form: host threads that keep creating predictors and growing their workspaces (every fresh workspace starts with
zero-filled rectangle-sum images and store masks) while the other threads' kernels run: the first batches of a new workspace must
come out right whatever else the device is doing."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W, H = 320, 240
forest = synth.fit_forest(8, 12, synth.FOREST_SEED_BASE + 81, n_frames=12, subset=2000)
model = synth.ModelParams(stepwidth=2)
frames = synth.biwi_batch(40, W, H, first=900)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
with HoughPrediction(forest, model) as hp0:
    ref = hp0.predict_batch(frames, intr).copy()
bad = [0] * n_threads
errs = []


def work(t):
    try:
        rs = np.random.RandomState(500 + t)
        for it in range(iters):
            with HoughPrediction(forest, model) as hp:
                for n in sorted(int(x) for x in rs.randint(1, 41, 3)):          # growing batches: three workspaces per predictor
                    idx = rs.randint(0, 40, n)
                    out = hp.predict_batch(frames[idx].copy(), intr)
                    if not (np.array_equal(out["mid_point"], ref["mid_point"][idx]) and np.array_equal(out["rotation"], ref["rotation"][idx])):
                        bad[t] += 1
                        errs.append((t, it, n))
    except Exception as e:   # noqa
        errs.append((t, repr(e)))


ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
[t.start() for t in ths]
[t.join() for t in ths]
print(n_threads, "threads x", iters, "predictors x 3 growing batches: mismatching per thread", bad, "errors", errs[:10])
