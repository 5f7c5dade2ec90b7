"""Experiment (GPU box): do consecutive batches overlap usefully when they alternate between two predictors on two
streams (the latency-bound tail kernels of batch i beside the bandwidth / issue-bound head kernels of batch i + 1)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

NF, W, H = 256, 640, 480
dev = torch.device("cuda:0")
forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
model = synth.ModelParams(stepwidth=4)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
base = synth.biwi_batch(64, W, H)
frames = torch.from_numpy(np.concatenate([base] * 4).view(np.int16)).to(dev)
for nstreams in (1, 2, 3):
    hps = [HoughPrediction(forest, model) for _ in range(nstreams)]
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    outs = [torch.zeros(NF * 40, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    for hp in hps:
        hp.reserve(NF, W, H)
    def run(steps):
        for i in range(steps):
            k = i % nstreams
            hps[k].predict_batch_device(frames.data_ptr(), NF, W, H, intr, outs[k].data_ptr(), stream=streams[k].cuda_stream)
    run(6); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{nstreams} stream(s): {60 * NF / dt:9.0f} frames/s   {dt / 60 * 1e3:.4f} ms per batch")
    ref = outs[0].cpu().numpy().tobytes()
    assert all(o.cpu().numpy().tobytes() == ref for o in outs)
    for hp in hps:
        hp.close()
