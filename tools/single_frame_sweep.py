import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from depthhead_amd import synth
from depthhead_amd._lib import POSE_DTYPE
from depthhead_amd.prediction import HoughPrediction, IntrinsicMatrix
a = sys.argv[1:]
W, H, stride = int(a[0]), int(a[1]), int(a[2])
combos = [c.split(":") for c in a[3:]]
forest = synth.fit_forest(10, 15, synth.FOREST_SEED_BASE + 2)
dev = torch.device("cuda:0")
fr = torch.from_numpy(synth.biwi_batch(1, W, H).view(np.int16)).to(dev)
intr = IntrinsicMatrix(synth.default_intrinsic(W, H))
out = torch.zeros(POSE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev)
ref = None
for tile, top in combos:
    env = {}
    if tile != "-": env["DH_TILE"] = tile
    if top != "-": env["DH_TOP_LEVELS"] = top
    os.environ.update(env)
    try:
        with HoughPrediction(forest, synth.ModelParams(stepwidth=stride), device=0) as hp:
            hp.reserve(1, W, H)
            for _ in range(5):
                hp.predict_batch_device(fr.data_ptr(), 1, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            pose = out.cpu().numpy().tobytes()
            if ref is None: ref = pose
            geo = hp.debug_geometry()
            hp.set_profiling(True)
            acc = {}
            for _ in range(40):
                hp.predict_batch_device(fr.data_ptr(), 1, W, H, intr, out.data_ptr(), stream=st.cuda_stream)
                for k, v in hp.timing().items():
                    acc[k] = acc.get(k, 0.0) + v / 40
        print(f"{W}x{H} s{stride} DH_TILE={tile} DH_TOP_LEVELS={top} -> tile {geo['px']}x{geo['py']} top {geo['top_levels']} boxsum {acc['boxsum_ms']*1e3:5.1f} traverse {acc['traverse_ms']*1e3:6.1f} emit {acc['emit_ms']*1e3:5.1f} total {acc['total_ms']*1e3:6.1f} us same_pose {pose == ref}")
    finally:
        for k in env: os.environ.pop(k, None)
