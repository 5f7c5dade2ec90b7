"""Single-frame latency: one HBM-resident frame per call, replayed from a hipGraph and launched directly; prints wall time per
frame.  Under `rocprofv3 --kernel-trace` the per-kernel start / end stamps show where the time between kernels goes.
    python tools/single_frame_trace.py [w h stride reps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd._lib import POSE_DTYPE
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

a = sys.argv[1:]
W, H, stride, reps = (int(a[i]) if len(a) > i else d for i, d in ((0, 320), (1, 240), (2, 1), (3, 200)))
forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
dev = torch.device("cuda:0")
fr = torch.from_numpy(synth.biwi_batch(1, W, H).view(np.int16)).to(dev)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
out = torch.zeros(POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev)
with HoughPrediction(forest, synth.ModelParams(stepwidth=stride), device=0) as hp:
    hp.reserve(1, W, H)
    for mode in ("direct", "graph"):
        if mode == "graph":
            hp.graph_capture(fr.data_ptr(), 1, W, H, intr, out.data_ptr())
        run = (lambda: hp.graph_launch(st.cuda_stream)) if mode == "graph" else (lambda: hp.predict_batch_device(fr.data_ptr(), 1, W, H, intr, out.data_ptr(), stream=st.cuda_stream))
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        # latency: one frame at a time, the host waits for each (the live-camera loop of examples/live_prediction.rs:76-86)
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
            torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / reps * 1e6
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        thr = (time.perf_counter() - t0) / reps * 1e6
        print(f"{W}x{H} stride {stride} {mode}: {lat:.1f} us per frame with a host sync after each, {thr:.1f} us back to back")
