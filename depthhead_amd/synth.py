"""Seeded synthetic inputs: BIWI-shaped depth frames and Hough forests.

Neither the BIWI data nor the released pretrained forest are available offline (Readme.md:17-45,
72-74 of the reference), so tests and the bench use these generators.
Everything is a pure function of the seed (counter-based splitmix64), independent of numpy's
global RNG, so the GPU box regenerates bit-identical inputs.

Geometry follows the only in-tree trainer (examples/hough_tree_trainer.rs:149-165): 80x80 patches,
split rectangles of 0.3 * patch = 24x24 with top-left in [0,56)^2 (src/types.rs:82-91), thresholds
uniform in [-256, 256) (src/hough/houghforest.rs:230-234), sigma 8, 20 mean-shift iterations
(src/hough/prediction.rs:225-233).
"""
from __future__ import annotations

import numpy as np

from .forest import NODE_DTYPE, Forest

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

FRAME_SEED_BASE = 0xD0E70000
FOREST_SEED_BASE = 0xF0BE5700


class SplitMix:
    """Vectorised splitmix64 stream."""

    def __init__(self, seed: int):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            z = self.state + np.arange(1, n + 1, dtype=np.uint64) * _G
            self.state = np.uint64(self.state + np.uint64(n) * _G)
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            return z ^ (z >> np.uint64(31))

    def uniform(self, n: int) -> np.ndarray:
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def randint(self, lo: int, hi: int, n: int) -> np.ndarray:
        """Integers in [lo, hi] inclusive."""
        return (self.u64(n) % np.uint64(hi - lo + 1)).astype(np.int64) + lo

    def normal(self, n: int) -> np.ndarray:
        u1 = 1.0 - self.uniform(n)
        u2 = self.uniform(n)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def default_intrinsic(w: int = 640, h: int = 480) -> np.ndarray:
    """`IntrinsicMatrix::default_kinect_intrinsic` (src/types.rs:418-420), scaled with the frame."""
    f = 560.0 * w / 640.0
    return np.array([[f, 0.0, w / 2.0], [0.0, f, h / 2.0], [0.0, 0.0, 1.0]], dtype=np.float32)


def head_truth(w: int, h: int, seed: int):
    """(z0, hx, hy, rotation_deg[3]) of the synthetic subject of `biwi_like(w, h, seed)`."""
    rng = SplitMix(seed)
    u = rng.uniform(3)
    z0 = 700.0 + 500.0 * u[0]
    hx = w * (0.25 + 0.5 * u[1])
    hy = h * (0.25 + 0.5 * u[2])
    rot = np.array([(hx - w / 2.0) / w * 80.0, (hy - h / 2.0) / h * 60.0, (z0 - 950.0) / 250.0 * 20.0])
    return z0, hx, hy, rot


def biwi_like(w: int = 640, h: int = 480, seed: int = FRAME_SEED_BASE) -> np.ndarray:
    """One Kinect-style uint16 depth frame (mm): zero background, a 95 mm sphere "head" at
    700-1200 mm, a torso slab below it, +-2 mm integer noise and 2 % zero holes."""
    z0, hx, hy, _ = head_truth(w, h, seed)
    rng = SplitMix(seed ^ 0x5EED5EED)
    fx = 560.0 * w / 640.0
    R = 95.0
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    dxm = (xx - hx) * z0 / fx
    dym = (yy - hy) * z0 / fx
    d2 = dxm * dxm + dym * dym
    depth = np.zeros((h, w), dtype=np.float64)
    head = d2 < R * R
    depth[head] = z0 - np.sqrt(R * R - d2[head])
    rp = R * fx / z0
    torso = (yy > hy + 0.9 * rp) & (np.abs(xx - hx) < 1.5 * rp) & ~head
    depth[torso] = z0 + 50.0
    fg = depth > 0
    noise = rng.randint(-2, 2, w * h).reshape(h, w)
    holes = rng.uniform(w * h).reshape(h, w) < 0.02
    out = np.where(fg, np.rint(depth) + noise, 0.0)
    out[holes] = 0.0
    return np.clip(out, 0, 65535).astype(np.uint16)


def biwi_batch(n: int, w: int = 640, h: int = 480, first: int = 0) -> np.ndarray:
    """Frames `first .. first+n-1` of the canonical stream (seed = FRAME_SEED_BASE + index)."""
    out = np.empty((n, h, w), dtype=np.uint16)
    for i in range(n):
        out[i] = biwi_like(w, h, FRAME_SEED_BASE + first + i)
    return out


def synth_forest(n_trees: int = 10, max_depth: int = 15, seed: int = FOREST_SEED_BASE,
                 patch: tuple[int, int] = (80, 80), rect_scale: float = 0.3,
                 full_depth: int = 6, p_split: float = 0.75, rect_scale_max: float | None = None) -> Forest:
    """Random forest with the trainer's geometry.

    Structure: nodes shallower than `full_depth` always split, deeper ones split with probability
    `p_split` until `max_depth` (the reference stops at depth >= max_depth,
    src/hough/houghforest.rs:306).  Nodes of a tree are numbered breadth-first and stored
    contiguously, so the top levels of every tree are one dense prefix.

    So that a realistic share of patches passes the mean-probability gate (prob > 0.7,
    src/hough/prediction.rs:584) the root of every tree tests "patch centre in front of the rows
    above it" and leaves below its `one` child are positive with p = 0.95, those below `zero` with
    p = 0.1.  Positive leaves carry 2..24 votes; the per-leaf spread varies so both covariance
    gates (prediction.rs:600, :643) fire and fail.
    """
    pw, ph = patch
    rw, rh = int(pw * rect_scale), int(ph * rect_scale)
    rng = SplitMix(seed)
    roots = np.zeros(n_trees, dtype=np.int32)
    node_chunks, leaf_side_chunks = [], []
    node_base = leaf_base = 0
    for t in range(n_trees):
        # ---- structure, level by level (slots = child positions still to fill)
        cz = np.zeros(0, dtype=np.int64); co = np.zeros(0, dtype=np.int64)   # per node (local, BFS order)
        leaf_side = []
        slot_parent = np.array([-1]); slot_which = np.array([0]); slot_side = np.array([0])
        n_nodes = n_leaves = 0
        root_val = 0
        for d in range(max_depth + 1):
            k = slot_parent.size
            if k == 0:
                break
            u = rng.uniform(k)
            split = np.full(k, d < max_depth) & ((d < full_depth) | (u < p_split))
            ns = int(split.sum()); nl = k - ns
            val = np.empty(k, dtype=np.int64)
            val[split] = node_base + n_nodes + np.arange(ns)
            val[~split] = ~(leaf_base + n_leaves + np.arange(nl))
            leaf_side.append(slot_side[~split])
            if d == 0:
                root_val = int(val[0])
            else:  # hook into parents
                sel = slot_which == 0
                cz[slot_parent[sel]] = val[sel]
                co[slot_parent[~sel]] = val[~sel]
            cz = np.concatenate([cz, np.zeros(ns, dtype=np.int64)])
            co = np.concatenate([co, np.zeros(ns, dtype=np.int64)])
            new_ids = n_nodes + np.arange(ns)
            n_nodes += ns; n_leaves += nl
            side_new = slot_side[split]
            slot_parent = np.repeat(new_ids, 2)
            slot_which = np.tile(np.array([0, 1]), ns)
            slot_side = np.repeat(side_new, 2) if d > 0 else np.tile(np.array([0, 1]), ns)
        roots[t] = root_val
        nodes = np.zeros(n_nodes, dtype=NODE_DTYPE)
        nodes["child_zero"] = cz.astype(np.int32)
        nodes["child_one"] = co.astype(np.int32)
        for key in ("r1", "r2"):
            if rect_scale_max is None:      # the trainer's geometry: one rectangle size for the whole forest
                w_, h_ = np.full(n_nodes, rw), np.full(n_nodes, rh)
            else:                           # mixed sizes (general path): scale drawn per rectangle
                sc = rect_scale + (rect_scale_max - rect_scale) * rng.uniform(n_nodes)
                w_, h_ = np.maximum((pw * sc).astype(np.int64), 1), np.maximum((ph * sc).astype(np.int64), 1)
            x0 = (rng.uniform(n_nodes) * (pw - w_)).astype(np.int64)     # Rect::scale_and_replace, types.rs:82-91
            y0 = (rng.uniform(n_nodes) * (ph - h_)).astype(np.int64)
            nodes[key] = np.stack([x0, y0, x0 + w_, y0 + h_], axis=1).astype(np.uint16)
        nodes["threshold"] = rng.uniform(n_nodes) * 512.0 - 256.0
        if n_nodes:
            cx0, cy0 = (pw - rw) // 2, (ph - rh) // 2
            nodes["r1"][0] = (cx0, cy0, cx0 + rw, cy0 + rh)
            tx = int(rng.randint(0, pw - rw - 1, 1)[0])
            ty = int(rng.randint(0, 7, 1)[0])
            nodes["r2"][0] = (tx, ty, tx + rw, ty + rh)
            nodes["threshold"][0] = 50.0 + 100.0 * rng.uniform(1)[0]
        node_chunks.append(nodes)
        leaf_side_chunks.append(np.concatenate(leaf_side) if leaf_side else np.zeros(0, dtype=np.int64))
        node_base += n_nodes; leaf_base += n_leaves

    nodes = np.concatenate(node_chunks) if node_chunks else np.zeros(0, dtype=NODE_DTYPE)
    side = np.concatenate(leaf_side_chunks)
    L = side.size
    positive = rng.uniform(L) < np.where(side == 1, 0.95, 0.1)
    n_votes = np.where(positive, rng.randint(2, 24, L), 0)
    m_neg = (rng.uniform(L) * (n_votes // 4 + 1)).astype(np.int64)
    prob = np.where(positive, n_votes / np.maximum(n_votes + m_neg, 1), 0.0)
    begin = np.zeros(L + 1, dtype=np.uint32)
    np.cumsum(n_votes, out=begin[1:])
    nv = int(begin[-1])
    leaf_of_vote = np.repeat(np.arange(L), n_votes)
    mu_off = (rng.uniform(L * 3).reshape(L, 3) * 300.0 - 150.0)
    sg_off = 10.0 + 50.0 * rng.uniform(L)
    mu_rot = (rng.uniform(L * 3).reshape(L, 3) * 120.0 - 60.0)
    sg_rot = 2.0 + 14.0 * rng.uniform(L)
    offsets = (mu_off[leaf_of_vote] + rng.normal(nv * 3).reshape(nv, 3) * sg_off[leaf_of_vote, None]).astype(np.float32)
    rotations = mu_rot[leaf_of_vote] + rng.normal(nv * 3).reshape(nv, 3) * sg_rot[leaf_of_vote, None]
    return Forest(roots, nodes, prob, begin, begin.copy(), offsets, rotations)


class ModelParams:
    """The serialised scalars of `HoughPrediction` (src/hough/prediction.rs:239-256)."""

    def __init__(self, stepwidth=10, subimage_width=80, subimage_height=80, gaussian_sigma=8.0,
                 meanshift_iterations=20):
        self.stepwidth = int(stepwidth)
        self.subimage_width = int(subimage_width)
        self.subimage_height = int(subimage_height)
        self.gaussian_sigma = float(gaussian_sigma)
        self.meanshift_iterations = int(meanshift_iterations)

    def patch_grid(self, w: int, h: int) -> tuple[int, int]:
        """Sliding-window positions (src/hough/prediction.rs:535-548, 684-686)."""
        lw = self.subimage_width // 2; rw = self.subimage_width - lw
        lh = self.subimage_height // 2; rh = self.subimage_height - lh
        nx = len(range(lw, w - rw, self.stepwidth)) if w >= self.subimage_width else 0
        ny = len(range(lh, h - rh, self.stepwidth)) if h >= self.subimage_height else 0
        return nx, ny


# ------------------------------------------------------------------ a forest fitted to synthetic subjects
TRAIN_SEED_BASE = 0x7A110000


def _frame_sat(img: np.ndarray) -> np.ndarray:
    s = np.zeros((img.shape[0] + 1, img.shape[1] + 1), dtype=np.int64)
    s[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    return s


def fit_forest(n_trees: int = 10, max_depth: int = 15, seed: int = FOREST_SEED_BASE, n_frames: int = 48,
               per_frame: int = 160, subset: int = 6000, n_candidates: int = 8, min_subset: int = 20,
               patch: tuple[int, int] = (80, 80), rect_scale: float = 0.3, w: int = 640, h: int = 480,
               steepness: float = 5.0, max_votes: int = 40) -> Forest:
    """A small Hough forest actually FITTED to synthetic subjects, so that its votes are coherent
    (leaves reached by head patches point at the head centre) and the vote / mean-shift stages see
    a realistic load.  This mirrors the shape of the reference's trainer -- patches labelled by a
    head mask, 3-D offset + rotation truths, random rectangle-pair features with thresholds in
    [-256, 256), best-of-N by an entropy + regression-uncertainty score, stop below `min_subset`
    samples / at `max_depth` / when no positive remains (src/hough/prediction.rs:145-234,
    src/hough/houghforest.rs:204-310) -- but it is a generator for benchmarks and tests, not a
    restatement: training is outside the scope of this repository.
    """
    pw, ph = patch
    rw, rh = int(pw * rect_scale), int(ph * rect_scale)
    lw, lh = pw // 2, ph // 2
    rng = SplitMix(seed ^ 0xF17F0E57)
    f = 560.0 * w / 640.0
    cx0, cy0 = w / 2.0, h / 2.0
    # ---- samples
    sats, S_frame, S_x, S_y, S_pos, S_off, S_rot = [], [], [], [], [], [], []
    gx, gy = np.arange(lw, w - (pw - lw), 4), np.arange(lh, h - (ph - lh), 4)
    GX, GY = np.meshgrid(gx, gy)
    GX, GY = GX.ravel(), GY.ravel()
    for i in range(n_frames):
        fseed = TRAIN_SEED_BASE + i
        img = biwi_like(w, h, fseed)
        z0, hx, hy, rot = head_truth(w, h, fseed)
        sat = _frame_sat(img)
        sats.append(sat)
        ox, oy = GX - lw, GY - lh
        psum = sat[oy + ph, ox + pw] - sat[oy, ox + pw] - sat[oy + ph, ox] + sat[oy, ox]
        nonbg = psum > 0                                                        # prediction.rs:181-203: only non-background patches
        zc = img[GY, GX].astype(np.float64)
        rp = 95.0 * f / z0
        inhead = ((GX - hx) ** 2 + (GY - hy) ** 2 < (0.85 * rp) ** 2) & (zc > 0)
        pos_idx = np.flatnonzero(nonbg & inhead)
        neg_idx = np.flatnonzero(nonbg & ~inhead)
        for idx, want, label in ((pos_idx, per_frame // 2, True), (neg_idx, per_frame // 2, False)):
            if idx.size == 0:
                continue
            pick = idx[(rng.uniform(min(want, idx.size)) * idx.size).astype(np.int64)]
            xs, ys, zs = GX[pick].astype(np.float64), GY[pick].astype(np.float64), zc[pick]
            p3 = np.stack([(xs - cx0) / f * zs, (ys - cy0) / f * zs, zs], axis=1)
            head3 = np.array([(hx - cx0) / f * z0, (hy - cy0) / f * z0, z0])
            S_frame.append(np.full(pick.size, i)); S_x.append(GX[pick] - lw); S_y.append(GY[pick] - lh)
            S_pos.append(np.full(pick.size, label)); S_off.append(p3 - head3); S_rot.append(np.tile(rot, (pick.size, 1)))
    S_frame, S_x, S_y = np.concatenate(S_frame), np.concatenate(S_x), np.concatenate(S_y)
    S_pos, S_off, S_rot = np.concatenate(S_pos), np.concatenate(S_off), np.concatenate(S_rot)
    SAT = np.stack(sats)
    n_samples = S_frame.size

    def rect_avg(sel, r):
        fr, x, y = S_frame[sel], S_x[sel], S_y[sel]
        s = SAT[fr, y + r[3], x + r[2]] - SAT[fr, y + r[1], x + r[2]] - SAT[fr, y + r[3], x + r[0]] + SAT[fr, y + r[1], x + r[0]]
        return s / float((r[2] - r[0]) * (r[3] - r[1]))

    def neg_entropy(pos):
        p = pos.mean() if pos.size else 0.0
        return (p * np.log(p) if p > 0 else 0.0) + ((1 - p) * np.log(1 - p) if p < 1 else 0.0)

    def reg_unc(sel):
        good = sel[S_pos[sel]]
        if good.size < 2:
            return 0.0
        return float(np.log(S_off[good].var(axis=0, ddof=1).sum() + S_rot[good].var(axis=0, ddof=1).sum() + 1.0))

    roots = np.zeros(n_trees, dtype=np.int32)
    all_nodes, leaf_prob, leaf_off, leaf_rot = [], [], [], []
    for t in range(n_trees):
        sub = (rng.uniform(min(subset, n_samples)) * n_samples).astype(np.int64)
        nodes_t = []                     # (r1, r2, thr, child_zero, child_one) with local indices, fixed up below
        frontier = [(sub, 0, -1, 0)]     # (samples, depth, parent, which)
        root_val = None
        while frontier:
            nxt = []
            for sel, depth, parent, which in frontier:
                pos = S_pos[sel]
                is_leaf = depth >= max_depth or sel.size < min_subset or not pos.any()     # houghforest.rs:302-310
                best = None
                if not is_leaf:
                    fac = 1.0 - np.exp(-depth / steepness)
                    for _ in range(n_candidates):
                        u = rng.uniform(5)
                        r1x, r1y = int(u[0] * (pw - rw)), int(u[1] * (ph - rh))
                        r2x, r2y = int(u[2] * (pw - rw)), int(u[3] * (ph - rh))
                        r1, r2 = (r1x, r1y, r1x + rw, r1y + rh), (r2x, r2y, r2x + rw, r2y + rh)
                        thr = u[4] * 512.0 - 256.0
                        one = rect_avg(sel, r1) - rect_avg(sel, r2) > thr
                        n1 = int(one.sum())
                        if n1 == 0 or n1 == sel.size:
                            continue
                        l, r = sel[~one], sel[one]
                        wl, wr = l.size / sel.size, r.size / sel.size
                        score = -(wl * neg_entropy(S_pos[l]) + wr * neg_entropy(S_pos[r])) + fac * (wl * reg_unc(l) + wr * reg_unc(r))
                        if best is None or score < best[0]:
                            best = (score, r1, r2, thr, l, r)
                    is_leaf = best is None
                if is_leaf:
                    good = sel[pos][:max_votes]
                    val = ~len(leaf_prob)
                    leaf_prob.append(pos.mean() if sel.size else 0.0)
                    leaf_off.append(S_off[good]); leaf_rot.append(S_rot[good])
                else:
                    val = len(all_nodes) + len(nodes_t)
                    nodes_t.append([best[1], best[2], best[3], 0, 0])
                    nxt.append((best[4], depth + 1, len(nodes_t) - 1, 0))
                    nxt.append((best[5], depth + 1, len(nodes_t) - 1, 1))
                if parent < 0:
                    root_val = val
                else:
                    nodes_t[parent][3 + which] = val
            frontier = nxt
        roots[t] = root_val
        all_nodes.extend(nodes_t)
    nodes = np.zeros(len(all_nodes), dtype=NODE_DTYPE)
    for i, (r1, r2, thr, cz, co) in enumerate(all_nodes):
        nodes[i] = (r1, r2, thr, cz, co)
    n_votes = np.array([len(o) for o in leaf_off], dtype=np.int64)
    begin = np.zeros(len(leaf_prob) + 1, dtype=np.uint32)
    np.cumsum(n_votes, out=begin[1:])
    offsets = np.concatenate(leaf_off).astype(np.float32) if n_votes.sum() else np.zeros((0, 3), dtype=np.float32)
    rotations = np.concatenate(leaf_rot).astype(np.float64) if n_votes.sum() else np.zeros((0, 3))
    return Forest(roots, nodes, np.array(leaf_prob), begin, begin.copy(), offsets, rotations)
