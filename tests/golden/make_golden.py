"""Regenerates tests/golden/*.npz: small inputs with the CPU oracle's outputs at every stage.

The reference (Rust) cannot be run in this pipeline and holds no fixture for this path, so these
are REGRESSION vectors produced by the oracle (oracle/dh_oracle.c), not reference outputs: they
freeze today's verified behaviour (tests/test_hand_case.py checks one case on paper, the KATs pin
the helpers) so the oracle itself cannot drift, and they let the GPU tests run without the oracle.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from depthhead_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

CASES = {
    # name: (w, h, n_frames, trees, depth, stride, forest seed offset, first frame, iterations)
    "tiny_96x96_s4": (96, 96, 3, 3, 5, 4, 201, 100, 20),
    "small_128x112_s3": (128, 112, 2, 5, 7, 3, 202, 110, 20),
    "small_160x120_s7_it5": (160, 120, 2, 4, 6, 7, 203, 120, 5),
}


def make(name):
    w, h, n, trees, depth, stride, fseed, first, iters = CASES[name]
    forest = synth.synth_forest(trees, depth, synth.FOREST_SEED_BASE + fseed)
    model = synth.ModelParams(stepwidth=stride, meanshift_iterations=iters)
    frames = synth.biwi_batch(n, w, h, first=first)
    K = synth.default_intrinsic(w, h)
    out = dict(frames=frames, K=K, params=np.array([stride, 80, 80, iters], dtype=np.int64), sigma=np.float32(8.0),
               roots=forest.roots, nodes=forest.nodes.view(np.uint8), leaf_prob=forest.leaf_prob,
               off_begin=forest.off_begin, rot_begin=forest.rot_begin, offsets=forest.offsets, rotations=forest.rotations)
    for i in range(n):
        r = po.predict(forest, model, frames[i], K, rect_mode=po.RECT_FAITHFUL)
        out[f"leaf_idx_{i}"] = r.leaf_idx
        out[f"patch_flags_{i}"] = r.patch_flags
        out[f"pos_grid_{i}"] = r.pos_grid
        nz = np.flatnonzero(r.rot_grid)
        out[f"rot_grid_nz_{i}"] = np.stack([nz, r.rot_grid[nz]], axis=1).astype(np.int64)
        out[f"guess_{i}"] = np.concatenate([r.guess_mid, r.guess_rot])
        out[f"mid_cells_{i}"] = r.mid_cells
        out[f"rot_cells_{i}"] = r.rot_cells
        out[f"ms_trace_mid_{i}"] = r.ms_trace_mid
        out[f"ms_trace_rot_{i}"] = r.ms_trace_rot
        out[f"mid_point_{i}"] = r.mid_point
        out[f"rotation_{i}"] = r.rotation
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")


if __name__ == "__main__":
    for c in CASES:
        make(c)
