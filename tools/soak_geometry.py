"""Soak, second form: ONE predictor, the frame size changes from call to call (every change reallocates the workspace and
rebuilds the geometry-specific node table), entry points mixed; every pose must equal the reference pass of that size."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import biwi, synth
from depthhead_amd._lib import POSE_DTYPE
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
forest = synth.fit_forest(8, 12, synth.FOREST_SEED_BASE + 81, n_frames=12, subset=2000)
sizes = [(640, 480), (320, 240), (400, 300), (644, 482), (200, 160)]
dev = torch.device("cuda:0")
rs = np.random.RandomState(9)
for stride in (4, 3):
    model = synth.ModelParams(stepwidth=stride)
    data = {}
    for (w, h) in sizes:
        fr = np.stack([synth.biwi_like(max(w, 96), max(h, 96), 7000 + i)[:h, :w] for i in range(24)]).copy()
        fr[3] = 0
        with HoughPrediction(forest, model) as hp0:
            intr = IntrinsicMatrix(synth.default_intrinsic(w, h))
            data[(w, h)] = (fr, intr, hp0.predict_batch(fr, intr).copy(), hp0.predict_mask(fr[:3]).copy(),
                            hp0.predict_parameter_from2dhough(fr[:3], intr).copy(), [biwi.encode_depth(f) for f in fr])
    bad = 0
    with HoughPrediction(forest, model) as hp:
        for it in range(iters):
            w, h = sizes[rs.randint(0, len(sizes))]
            fr, intr, ref, rmask, r2d, pay = data[(w, h)]
            n = int(rs.randint(1, 25))
            idx = rs.randint(0, 24, n)
            kind = int(rs.randint(0, 6))
            if kind == 0:
                out = hp.predict_batch(fr[idx].copy(), intr); ok = np.array_equal(out["mid_point"], ref["mid_point"][idx]) and np.array_equal(out["rotation"], ref["rotation"][idx])
            elif kind == 1:
                out = hp.predict_batch_rle([pay[i] for i in idx], intr); ok = np.array_equal(out["mid_point"], ref["mid_point"][idx]) and np.array_equal(out["rotation"], ref["rotation"][idx])
            elif kind == 2:
                ok = np.array_equal(hp.predict_mask(fr[:3]), rmask)
            elif kind == 3:
                ok = np.array_equal(hp.predict_parameter_from2dhough(fr[:3], intr)["mid_point"], r2d["mid_point"])
            elif kind == 4:
                d = torch.from_numpy(fr[idx].view(np.int16)).to(dev); o = torch.zeros(n * 40, dtype=torch.uint8, device=dev)
                hp.predict_batch_device(d.data_ptr(), n, w, h, intr, o.data_ptr()); torch.cuda.synchronize()
                out = np.frombuffer(o.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
                ok = np.array_equal(out["mid_point"], ref["mid_point"][idx]) and np.array_equal(out["rotation"], ref["rotation"][idx])
            else:
                d = torch.from_numpy(fr[idx[:2].repeat(2)[:2]].view(np.int16)).to(dev); o = torch.zeros(2 * 40, dtype=torch.uint8, device=dev)
                hp.graph_capture(d.data_ptr(), 2, w, h, intr, o.data_ptr())
                st = torch.cuda.current_stream(dev)
                ok = True
                for _ in range(3):
                    j = rs.randint(0, 24, 2)
                    d.copy_(torch.from_numpy(fr[j].view(np.int16)))
                    hp.graph_launch(st.cuda_stream); st.synchronize()
                    out = np.frombuffer(o.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
                    ok = ok and np.array_equal(out["mid_point"], ref["mid_point"][j]) and np.array_equal(out["rotation"], ref["rotation"][j])
            if not ok:
                bad += 1
                print("MISMATCH stride", stride, "iteration", it, "size", (w, h), "kind", kind, "n", n, flush=True)
    print("stride", stride, ":", iters, "iterations, mismatching:", bad, flush=True)
