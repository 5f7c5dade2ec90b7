"""Flat Hough-forest container shared by the HIP path, the tests and the bench.

The reference stores the forest as `stamm::randforest::RandomForest<LeafParam, HoughTreeFunctions>`
(src/hough/prediction.rs:33-34) with `NodeParam{r1, r2, threshold}` (src/hough/houghforest.rs:63-68)
and `LeafParam{prob, offsets, rotations}` (src/hough/houghforest.rs:73-78).  Here the same data is
flattened into C-layout arrays that `dh_forest_create` (include/depthhead_hip.h) copies to HBM:

* ``roots[t]``      int32, node index of tree ``t`` (or ``~leaf`` when the tree is a single leaf)
* ``nodes``         32-byte records, see ``NODE_DTYPE`` -- both children are explicit
                    (``child_one`` is taken when ``avg(r1) - avg(r2) > threshold``), so the
                    stamm ``Binar`` -> child convention is fixed by whoever builds the arrays
* ``leaf_prob``     float64 per leaf
* ``off_begin`` / ``rot_begin``  CSR offsets (n_leaves + 1) into ``offsets`` (float32 x3, mm)
                    and ``rotations`` (float64 x3, degrees)
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

# x0, y0, x1, y1 (topleft / bottomright, src/types.rs:33-37) relative to the patch origin
NODE_DTYPE = np.dtype(
    [
        ("r1", "<u2", (4,)),
        ("r2", "<u2", (4,)),
        ("threshold", "<f8"),
        ("child_zero", "<i4"),
        ("child_one", "<i4"),
    ],
    align=True,
)
assert NODE_DTYPE.itemsize == 32


@dataclass
class Forest:
    roots: np.ndarray        # int32 [n_trees]
    nodes: np.ndarray        # NODE_DTYPE [n_nodes]
    leaf_prob: np.ndarray    # float64 [n_leaves]
    off_begin: np.ndarray    # uint32 [n_leaves + 1]
    rot_begin: np.ndarray    # uint32 [n_leaves + 1]
    offsets: np.ndarray      # float32 [n_off, 3]
    rotations: np.ndarray    # float64 [n_rot, 3]

    def __post_init__(self):
        self.roots = np.ascontiguousarray(self.roots, dtype=np.int32)
        self.nodes = np.ascontiguousarray(self.nodes, dtype=NODE_DTYPE)
        self.leaf_prob = np.ascontiguousarray(self.leaf_prob, dtype=np.float64)
        self.off_begin = np.ascontiguousarray(self.off_begin, dtype=np.uint32)
        self.rot_begin = np.ascontiguousarray(self.rot_begin, dtype=np.uint32)
        self.offsets = np.ascontiguousarray(self.offsets, dtype=np.float32).reshape(-1, 3)
        self.rotations = np.ascontiguousarray(self.rotations, dtype=np.float64).reshape(-1, 3)

    @property
    def n_trees(self) -> int:
        return int(self.roots.shape[0])

    @property
    def n_nodes(self) -> int:
        return int(self.nodes.shape[0])

    @property
    def n_leaves(self) -> int:
        return int(self.leaf_prob.shape[0])

    def nbytes(self) -> int:
        return sum(a.nbytes for a in (self.roots, self.nodes, self.leaf_prob, self.off_begin,
                                      self.rot_begin, self.offsets, self.rotations))

    def max_depth(self) -> int:
        """Longest root->leaf path in split nodes (iterative, no recursion limit)."""
        best = 0
        cz, co = self.nodes["child_zero"], self.nodes["child_one"]
        for r in self.roots:
            if r < 0:
                continue
            frontier = np.array([r], dtype=np.int64)
            depth = 0
            while frontier.size:
                depth += 1
                nxt = np.concatenate([cz[frontier], co[frontier]])
                frontier = nxt[nxt >= 0].astype(np.int64)
            best = max(best, depth)
        return best

    def save(self, path: str) -> None:
        np.savez_compressed(path, roots=self.roots, nodes=self.nodes.view(np.uint8), leaf_prob=self.leaf_prob,
                            off_begin=self.off_begin, rot_begin=self.rot_begin, offsets=self.offsets,
                            rotations=self.rotations)

    @staticmethod
    def load(path: str) -> "Forest":
        z = np.load(path, allow_pickle=False)
        return Forest(z["roots"], z["nodes"].view(NODE_DTYPE), z["leaf_prob"], z["off_begin"], z["rot_begin"],
                      z["offsets"], z["rotations"])


def ptr(a: np.ndarray, ctype):
    """ctypes pointer to a contiguous numpy array (kept alive by the caller)."""
    return a.ctypes.data_as(ctypes.POINTER(ctype))
