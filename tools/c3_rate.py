"""BASELINE configs[2] (50 trees, depth 20, stride 2, 32 frames of 640x480; bench.py's `c3_50x20_s2_32f` leg): frames/s by host clock
and per-kernel HIP-event times; with the profiling twin (DH_LIB_PATH=...knobs.so DH_CL_STOP=n) the phases of k_cluster."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from depthhead_amd import synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream(dev)
W, H = 640, 480
base = synth.biwi_batch(64, W, H)
fr = torch.from_numpy(np.concatenate([base] * 4).view(np.int16)).to(dev)
c3 = synth.synth_forest(50, 20, synth.FOREST_SEED_BASE + 3)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
with HoughPrediction(c3, synth.ModelParams(stepwidth=2), device=0) as hp:
    hp.reserve(32, W, H)
    v, _ = bench.rate(hp, fr, 32, W, H, intr, stream, steps=5, warmup=2)
    hp.set_profiling(True)
    out = torch.zeros(32 * 40, dtype=torch.uint8, device=dev)
    acc = {}
    for i in range(4):
        hp.predict_batch_device(fr.data_ptr(), 32, W, H, intr, out.data_ptr(), stream=stream.cuda_stream)
        for k, t in hp.timing().items():
            acc[k] = acc.get(k, 0.0) + t / 4
    print(os.environ.get("DH_CL_STOP", "full"), v, "frames/s", {k: round(t, 4) for k, t in acc.items()}, "hits/frame", int(hp.debug_hit_counts(32).mean()))
