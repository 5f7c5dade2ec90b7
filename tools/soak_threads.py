"""Soak, third form: host threads, each with its own predictor (a dh_predictor is single-threaded like the reference's
`HoughPrediction: !Sync`; distinct predictors may run concurrently), hammering different entry points at once.

On a mismatch the failing operation diagnoses itself before anything else runs on that predictor (round 2 saw ONE wrong
pose in 18 000 operations and could not say which stage produced it): the product-mode taps of the failing call (hit
counts, both guess grids) are compared with a quiet single-threaded reference pass of the same frames, and the call is
repeated on the same predictor (state kept) and on a fresh one.  That tells input (upload / decode) from rectangle sums /
walks (hit counts differ) from votes (grids differ) from mean shift (only the pose differs), and a persistent fault of the
predictor's state (the repeat is wrong again) from a transient one."""
import json, os, sys, threading
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
    ref_hits = hp0.debug_hit_counts(40).copy()
    ref_pos, ref_rot = [g.copy() for g in hp0.debug_grids(40)]
    rmask = hp0.predict_mask(frames[:3]).copy()
bad = [0] * n_threads
errs = []
quiet = threading.Lock()


def same(out, idx):
    return np.array_equal(out["mid_point"], ref["mid_point"][idx]) and np.array_equal(out["rotation"], ref["rotation"][idx])


def diagnose(hp, t, it, kind, idx, out):
    """Called with the other threads still running: first the taps of the failing call (still in the workspace), then repeats."""
    n = len(idx)
    rep = {"thread": t, "iteration": it, "op": kind, "n": n, "frames": [int(i) for i in idx]}
    wrong = [j for j in range(n) if not same(out[j:j + 1], idx[j:j + 1])]
    rep["wrong_slots"] = wrong
    rep["wrong_poses"] = [[out["mid_point"][j].tolist(), out["rotation"][j].tolist(), ref["mid_point"][idx[j]].tolist(), ref["rotation"][idx[j]].tolist()] for j in wrong[:4]]
    try:   # the taps describe the last device batch of the call (the whole call when it was not chunked: n < 24 here)
        ln = n if n < 24 else None
        if ln:
            hits = hp.debug_hit_counts(ln)
            pos, rot = hp.debug_grids(ln)
            rep["hit_counts_differ"] = [j for j in range(ln) if hits[j] != ref_hits[idx[j]]]
            rep["pos_grid_differs"] = [j for j in range(ln) if not np.array_equal(pos[j], ref_pos[idx[j]])]
            rep["rot_grid_differs"] = [j for j in range(ln) if not np.array_equal(rot[j], ref_rot[idx[j]])]
    except Exception as e:   # noqa
        rep["taps_error"] = repr(e)
    again = hp.predict_batch(frames[idx].copy(), intr) if kind == "predict_batch" else hp.predict_batch_rle([pay[i] for i in idx], intr)
    rep["repeat_on_same_predictor_ok"] = bool(same(again, idx))
    with quiet:
        with HoughPrediction(forest, model) as fresh:
            rep["fresh_predictor_ok"] = bool(same(fresh.predict_batch(frames[idx].copy(), intr), idx))
    return rep


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
                        errs.append({"thread": t, "iteration": it, "op": "predict_mask"})
                    continue
                if not same(out, idx):
                    bad[t] += 1
                    errs.append(diagnose(hp, t, it, "predict_batch" if k == 0 else "predict_batch_rle", idx, out))
    except Exception as e:   # noqa
        errs.append({"thread": t, "exception": repr(e)})


ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
[t.start() for t in ths]
[t.join() for t in ths]
print(n_threads, "threads x", iters, "iterations: mismatching per thread", bad, "errors", json.dumps(errs))
sys.exit(1 if errs else 0)
