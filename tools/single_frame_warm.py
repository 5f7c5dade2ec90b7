"""Is a single frame's latency a matter of GPU clocks?  Per-kernel HIP-event times of one 320x240 / 640x480 frame (a) after an idle
gap (host sync + sleep), (b) back to back, (c) launched right behind a 256-frame batch of another predictor on the same stream
(the chip busy and at full clock when the frame's kernels start).  GPU box, repo root: python tools/single_frame_warm.py [w h stride]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd._lib import POSE_DTYPE
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix
a = sys.argv[1:]
W, H, stride = (int(a[i]) if len(a) > i else d for i, d in ((0, 320), (1, 240), (2, 1)))
forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
dev = torch.device("cuda:0")
fr = torch.from_numpy(synth.biwi_batch(1, W, H).view(np.int16)).to(dev)
big = torch.from_numpy(synth.biwi_batch(64, 640, 480).view(np.int16)).to(dev)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H)); intr_big = IntrinsicMatrix(synth.default_intrinsic(640, 480))
out = torch.zeros(POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev); out_big = torch.zeros(64 * POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev)
with HoughPrediction(forest, synth.ModelParams(stepwidth=stride), device=0) as hp, HoughPrediction(forest, synth.ModelParams(stepwidth=4), device=0) as hb:
    hp.reserve(1, W, H); hb.reserve(64, 640, 480)
    for _ in range(5):
        hp.predict_batch_device(fr.data_ptr(), 1, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
        hb.predict_batch_device(big.data_ptr(), 64, 640, 480, intr_big, out_big.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    hp.set_profiling(True)
    for mode in ("idle 5 ms before", "back to back", "behind a 64-frame batch"):
        acc = {}
        reps = 30
        for _ in range(reps):
            if mode.startswith("idle"):
                torch.cuda.synchronize(); time.sleep(0.005)
            elif mode.startswith("behind"):
                hb.predict_batch_device(big.data_ptr(), 64, 640, 480, intr_big, out_big.data_ptr(), stream=st.cuda_stream)
            hp.predict_batch_device(fr.data_ptr(), 1, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
            for k, v in hp.timing().items():
                acc[k] = acc.get(k, 0.0) + v / reps
        print(f"{W}x{H} s{stride} {mode:26s} " + " ".join(f"{k[:-3]} {v * 1e3:6.1f}" for k, v in acc.items() if k.endswith("_ms")) + " us")
