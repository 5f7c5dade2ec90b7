"""Where a single frame's latency goes: per-kernel HIP-event times of one HBM-resident frame per call, full and with the
profiling twin cutting a kernel short after phase N (results of truncated runs are INVALID by construction).
    DH_LIB_PATH=depthhead_amd/libdepthhead_hip_knobs.so python tools/single_frame_phases.py [w h stride reps]
GPU box, repo root."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depthhead_amd import synth
from depthhead_amd._lib import POSE_DTYPE
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix

a = sys.argv[1:]
W, H, stride, reps = (int(a[i]) if len(a) > i else d for i, d in ((0, 320), (1, 240), (2, 1), (3, 50)))
forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
dev = torch.device("cuda:0")
fr = torch.from_numpy(synth.biwi_batch(1, W, H).view(np.int16)).to(dev)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
out = torch.zeros(POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev)
cuts = [("full", {})] + [(f"{k}={v}", {k: str(v)}) for k, vs in (("DH_TRAV_STOP", (9, 1, 3)), ("DH_EMIT_STOP", (1, 2)), ("DH_VOTE_STOP", (1, 2, 3)), ("DH_CL_STOP", (1, 2, 3))) for v in vs]
for name, env in cuts:
    os.environ.update(env)
    try:
        with HoughPrediction(forest, synth.ModelParams(stepwidth=stride), device=0) as hp:
            hp.reserve(1, W, H)
            for _ in range(5):
                hp.predict_batch_device(fr.data_ptr(), 1, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
            geo = hp.debug_geometry()
            hp.set_profiling(True)
            acc = {}
            for _ in range(reps):
                hp.predict_batch_device(fr.data_ptr(), 1, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
                for k, v in hp.timing().items():
                    acc[k] = acc.get(k, 0.0) + v / reps
        print(f"{W}x{H} s{stride} tile {geo['px']}x{geo['py']} top {geo['top_levels']} {name:16s} " + " ".join(f"{k[:-3]} {v * 1e3:6.1f}" for k, v in acc.items() if k.endswith("_ms")) + " us")
    finally:
        for k in env:
            os.environ.pop(k, None)
