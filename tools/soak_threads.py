"""Soak, third form: host threads, each with its own predictor (a dh_predictor is single-threaded like the reference's
`HoughPrediction: !Sync`; distinct predictors may run concurrently), hammering different entry points at once."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import biwi, synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W, H = 320, 240
forest = synth.fit_forest(8, 12, synth.FOREST_SEED_BASE + 81, n_frames=12, subset=2000)
model = synth.ModelParams(stepwidth=2)
frames = synth.biwi_batch(40, W, H, first=900)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
pay = [biwi.encode_depth(f) for f in frames]
with HoughPrediction(forest, model) as hp0:
    ref = hp0.predict_batch(frames, intr).copy()
    rmask = hp0.predict_mask(frames[:3]).copy()
bad = [0] * n_threads
errs = []

def work(t):
    try:
        rs = np.random.RandomState(100 + t)
        with HoughPrediction(forest, model) as hp:
            for it in range(iters):
                idx = rs.randint(0, 40, int(rs.randint(1, 41)))
                k = (it + t) % 3
                if k == 0:
                    out = hp.predict_batch(frames[idx].copy(), intr)
                elif k == 1:
                    out = hp.predict_batch_rle([pay[i] for i in idx], intr)
                else:
                    if not np.array_equal(hp.predict_mask(frames[:3]), rmask):
                        bad[t] += 1
                        errs.append((t, it, "predict_mask"))
                    continue
                if not (np.array_equal(out["mid_point"], ref["mid_point"][idx]) and np.array_equal(out["rotation"], ref["rotation"][idx])):
                    bad[t] += 1
                    wrong = [int(j) for j in range(len(idx)) if not (np.array_equal(out["mid_point"][j], ref["mid_point"][idx[j]]) and np.array_equal(out["rotation"][j], ref["rotation"][idx[j]]))]
                    errs.append((t, it, "predict_batch" if k == 0 else "predict_batch_rle", len(idx), wrong[:8], [int(idx[j]) for j in wrong[:8]]))
    except Exception as e:   # noqa
        errs.append((t, repr(e)))

ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
[t.start() for t in ths]
[t.join() for t in ths]
print(n_threads, "threads x", iters, "iterations: mismatching per thread", bad, "errors", errs)
