"""Loader for the committed fixtures under tests/golden/ (see make_golden.py)."""
import glob
import os

import numpy as np

from depthhead_amd import synth
from depthhead_amd.forest import Forest, NODE_DTYPE

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    forest = Forest(z["roots"], z["nodes"].view(NODE_DTYPE), z["leaf_prob"], z["off_begin"], z["rot_begin"],
                    z["offsets"], z["rotations"])
    stride, sw, sh, iters = (int(v) for v in z["params"])
    model = synth.ModelParams(stepwidth=stride, subimage_width=sw, subimage_height=sh,
                              gaussian_sigma=float(z["sigma"]), meanshift_iterations=iters)
    frames, K = z["frames"], z["K"]
    exp = []
    for i in range(frames.shape[0]):
        rot_grid = np.zeros(8000, dtype=np.uint32)
        nz = z[f"rot_grid_nz_{i}"]
        rot_grid[nz[:, 0]] = nz[:, 1].astype(np.uint32)
        exp.append(dict(leaf_idx=z[f"leaf_idx_{i}"], patch_flags=z[f"patch_flags_{i}"], pos_grid=z[f"pos_grid_{i}"],
                        rot_grid=rot_grid, guess=z[f"guess_{i}"], mid_cells=z[f"mid_cells_{i}"],
                        rot_cells=z[f"rot_cells_{i}"], ms_trace_mid=z[f"ms_trace_mid_{i}"],
                        ms_trace_rot=z[f"ms_trace_rot_{i}"], mid_point=z[f"mid_point_{i}"], rotation=z[f"rotation_{i}"]))
    return forest, model, frames, K, exp
