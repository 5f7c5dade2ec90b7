"""Soak: many batches of varying size / offset through ONE predictor, mixing every entry point (host frames pageable and
page-locked, run-length coded payloads, device-resident batches small and >= 512 (forked halves), hipGraph replay,
predict_mask, the 2-D Hough variant) and a second predictor running concurrently on another stream; every pose must equal
the pose the same frame got in a reference pass (catches stale per-batch state: counters, tile flags, window lists,
staging buffers, pre-gathered regions, captured graphs)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import biwi, synth
from depthhead_amd._lib import POSE_DTYPE, pinned_empty, DepthheadError
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W, H = 640, 480
forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
model = synth.ModelParams(stepwidth=4)
frames = synth.biwi_batch(96)
frames[5] = 0
frames[17, :, :320] = 0
payloads = [biwi.encode_depth(f) for f in frames]
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
dev = torch.device("cuda:0")
rs = np.random.RandomState(3)
bad = 0
with HoughPrediction(forest, model, device=0) as hp, HoughPrediction(forest, model, device=0) as hp2:
    ref = hp.predict_batch(frames, intr).copy()
    ref2d = hp.predict_parameter_from2dhough(frames[:8], intr).copy()
    refmask = hp.predict_mask(frames[:4]).copy()
    side = torch.cuda.Stream(dev)
    big = torch.from_numpy(np.concatenate([frames] * 6)[:520].view(np.int16)).to(dev)      # 520 frames: two forked halves
    big_out = torch.zeros(520 * 40, dtype=torch.uint8, device=dev)
    other = torch.from_numpy(frames.view(np.int16)).to(dev)
    other_out = torch.zeros(96 * 40, dtype=torch.uint8, device=dev)
    pin = pinned_empty((96, H, W), np.uint16)
    gfr = torch.from_numpy(frames[:2].view(np.int16)).to(dev).clone()
    gout = torch.zeros(2 * 40, dtype=torch.uint8, device=dev)
    have_graph = False

    def same(out, idx):
        return np.array_equal(out["mid_point"], ref["mid_point"][idx]) and np.array_equal(out["rotation"], ref["rotation"][idx])

    for it in range(iters):
        n = int(rs.randint(1, 97))
        idx = rs.randint(0, 96, n)
        kind = int(rs.randint(0, 8))
        # a second predictor works on its own stream all the while
        hp2.predict_batch_device(other.data_ptr(), 96, W, H, intr, other_out.data_ptr(), stream=side.cuda_stream)
        ok = True
        if kind == 0:
            ok = same(hp.predict_batch(frames[idx].copy(), intr), idx)
        elif kind == 1:
            pin[:n] = frames[idx]
            ok = same(hp.predict_batch(pin[:n], intr), idx)
        elif kind == 2:
            ok = same(hp.predict_batch_rle([payloads[i] for i in idx], intr), idx)
        elif kind == 3:
            hp.predict_batch_device(big.data_ptr(), 520, W, H, intr, big_out.data_ptr())
            torch.cuda.synchronize()
            out = np.frombuffer(big_out.cpu().numpy().tobytes(), dtype=POSE_DTYPE)
            ok = same(out, np.arange(520) % 96)
        elif kind == 4:
            ok = np.array_equal(hp.predict_parameter_from2dhough(frames[:8], intr)["mid_point"], ref2d["mid_point"])
        elif kind == 5:
            ok = np.array_equal(hp.predict_mask(frames[:4]), refmask)
        elif kind == 6:
            st = torch.cuda.current_stream(dev)
            if not have_graph:
                hp.graph_capture(gfr.data_ptr(), 2, W, H, intr, gout.data_ptr())
                have_graph = True
            j = rs.randint(0, 96, 2)
            gfr.copy_(torch.from_numpy(frames[j].view(np.int16)))
            try:
                hp.graph_launch(st.cuda_stream)
                st.synchronize()
                ok = same(np.frombuffer(gout.cpu().numpy().tobytes(), dtype=POSE_DTYPE), j)
            except DepthheadError as e:                  # a larger batch in between reallocated the workspace: stale by design
                assert e.code == -6, e
                have_graph = False
        else:
            d = torch.from_numpy(frames[idx].view(np.int16)).to(dev)
            o = torch.zeros(n * 40, dtype=torch.uint8, device=dev)
            hp.predict_batch_device(d.data_ptr(), n, W, H, intr, o.data_ptr())
            torch.cuda.synchronize()
            ok = same(np.frombuffer(o.cpu().numpy().tobytes(), dtype=POSE_DTYPE), idx)
        side.synchronize()
        ok2 = same(np.frombuffer(other_out.cpu().numpy().tobytes(), dtype=POSE_DTYPE), np.arange(96))
        if not (ok and ok2):
            bad += 1
            print("MISMATCH at iteration", it, "kind", kind, "n", n, "concurrent predictor ok:", ok2, flush=True)
        if it % 50 == 0:
            print("... iteration", it, "mismatches", bad, flush=True)
print("soak done,", iters, "iterations, mismatching batches:", bad)
