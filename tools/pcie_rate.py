"""Host-entry-point rates (GPU box): dh_predict_batch on pageable frames and dh_predict_batch_rle on BIWI payloads for a
few upload chunk sizes.  Usage: python tools/pcie_rate.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import biwi, synth
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 640, 480
forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
model = synth.ModelParams(stepwidth=4)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
distinct = synth.biwi_batch(64, W, H)
frames = np.concatenate([distinct] * ((n + 63) // 64))[:n]
enc = [biwi.encode_depth(f) for f in distinct]
payloads = [enc[i % 64] for i in range(n)]
print(f"{n} frames, {frames.nbytes / 1e6:.0f} MB raw, {sum(map(len, payloads)) / 1e6:.0f} MB run-length coded")
from depthhead_amd._lib import pinned_empty
pinned = pinned_empty(frames.shape, np.uint16)
pinned[...] = frames
for chunk in (16, 32, 64, 256):
    os.environ["DH_STAGE_CHUNK"] = str(chunk)
    with HoughPrediction(forest, model) as hp:
        ref = hp.predict_batch(frames, intr)
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter(); hp.predict_batch(frames, intr); best = min(best, time.perf_counter() - t0)
        bestp = 1e9
        for _ in range(6):
            t0 = time.perf_counter(); rp = hp.predict_batch(pinned, intr); bestp = min(bestp, time.perf_counter() - t0)
        assert rp.tobytes() == ref.tobytes()
        assert hp.predict_batch_rle(payloads, intr).tobytes() == ref.tobytes()
        bestr = 1e9
        for _ in range(6):
            t0 = time.perf_counter(); hp.predict_batch_rle(payloads, intr); bestr = min(bestr, time.perf_counter() - t0)
    print(f"chunk {chunk:4d}: pageable {n / best:9.0f} frames/s ({frames.nbytes / best / 1e9:5.1f} GB/s)   pinned {n / bestp:9.0f} frames/s ({frames.nbytes / bestp / 1e9:5.1f} GB/s)   rle {n / bestr:9.0f} frames/s")
