"""Per-kernel durations (HIP events, one batch in flight) of a workload: forest kind / size, stride, frame size, batch.
    python tools/kernel_times.py [fitted|synth] [trees] [depth] [stride] [w] [h] [frames] [reps]
Environment knobs (DH_*) apply as usual.  GPU box, repo root."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd._lib import POSE_DTYPE
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

a = sys.argv[1:]
kind = a[0] if len(a) > 0 else "fitted"
trees, depth, stride = (int(a[i]) if len(a) > i else d for i, d in ((1, 10), (2, 15), (3, 4)))
W, H, NF, reps = (int(a[i]) if len(a) > i else d for i, d in ((4, 640), (5, 480), (6, 256), (7, 10)))
rs = os.environ.get("KT_RECT_SCALE")          # synth only: rectangle edge as a fraction of the patch (0.3 = the trainer's 24 x 24)
forest = (synth.synth_forest(trees, depth, synth.FOREST_SEED_BASE + 2, rect_scale=float(rs)) if rs and kind != "fitted"
          else (synth.fit_forest if kind == "fitted" else synth.synth_forest)(trees, depth, synth.FOREST_SEED_BASE + 2))
dev = torch.device("cuda:0")
nd = min(64, NF)
fa = np.concatenate([synth.biwi_batch(nd, W, H)] * ((NF + nd - 1) // nd))[:NF]
fb = np.roll(np.concatenate([synth.biwi_batch(nd, W, H, first=10000)] * ((NF + nd - 1) // nd))[:NF], 7, axis=0)
batches = [torch.from_numpy(x.view(np.int16)).to(dev) for x in (fa, fb)]
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
out = torch.zeros(NF * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev)
with HoughPrediction(forest, synth.ModelParams(stepwidth=stride), device=0) as hp:
    hp.reserve(NF, W, H)
    for i in range(4):
        hp.predict_batch_device(batches[i % 2].data_ptr(), NF, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
    geo = hp.debug_geometry()
    hp.set_profiling(True)
    acc = {}
    for i in range(reps):
        hp.predict_batch_device(batches[i % 2].data_ptr(), NF, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
        for k, v in hp.timing().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    hits = hp.debug_hit_counts(NF)
print(f"{kind} {trees}x{depth} ({forest.n_nodes} nodes, {forest.n_leaves} leaves) stride {stride} {W}x{H} x{NF}: tile {geo['px']}x{geo['py']} top_levels {geo['top_levels']} "
      f"| " + " ".join(f"{k[:-3]} {v:.4f}" for k, v in acc.items() if k.endswith("_ms")) + f" | {NF / acc['total_ms'] / 1e3 * 1e3:.0f} k frames/s alone | hits/frame {hits.mean():.0f}")
