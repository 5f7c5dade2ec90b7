"""pyref.py -- a SECOND, independent CPU restatement of the hot path, written from the Rust text of the reference
(/root/reference/src/hough/prediction.rs:421-753, src/meanshift.rs:228-407, src/types.rs:317-339 and :424-445,
src/meancov_estimation.rs:201-216, :339-378, src/hough/houghforest.rs:185-193) and NOT from oracle/dh_oracle.c.

TEST INFRASTRUCTURE ONLY (tests/test_pyref.py): nothing under depthhead_amd/ imports it.  PARITY UNPINNED like the C
oracle -- the reference holds no golden vector for these stages and cannot be run here (Rust, no toolchain; the tree walk
lives in the un-vendored crate stamm 0.2.0).  What this file adds is a defence against a shared misreading: two restatements
by different routes (plain Python objects, a dict for the HashMap, numpy scalars for the f32 / f64 arithmetic, loops in
the reference's order) have to agree on every intermediate of the committed goldens and of the paper case.

Deliberately naive and slow (small frames only).  Conventions: `as` casts are Rust's (truncate toward zero, saturate,
NaN -> 0); f32 / f64 expressions are evaluated operation by operation on numpy scalars (no fused multiply-add);
u32 accumulators wrap; the flat forest format names both children, `child_one` being taken for Binar::One.
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

F32, F64 = np.float32, np.float64
ZSCALEFACTOR = 1            # prediction.rs:271
GUESS_GRID_PARTS = 20       # :276
ROT_GRID_PARTS = 120        # :280
MAX_VARIANCE_ROT = F64(400.0)       # :284
MAX_VARIANCE_OFFSET = F32(5200.0)   # :287

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]


def expf(x) -> np.float32:
    """f32::exp is the platform's expf."""
    return F32(_libm.expf(float(x)))


# ---------------------------------------------------------------- Rust `as`
def as_i32(v) -> int:
    v = float(v)
    if v != v:
        return 0
    if v >= 2147483648.0:
        return 2147483647
    if v <= -2147483648.0:
        return -2147483648
    return int(v)          # Python's int() truncates toward zero


def as_usize(v) -> int:
    v = float(v)
    if v != v or v <= 0.0:
        return 0
    if v >= 18446744073709551616.0:
        return 18446744073709551615
    return int(v)


def wrap_i32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


# ---------------------------------------------------------------- meancov_estimation.rs
def mat_vec(m, v, T):
    """impl Mul<Vec> for Mat (:201-216): tmp = rhs[0] * m[j][0]; tmp = tmp + rhs[i] * m[j][i]."""
    out = []
    for j in range(3):
        tmp = T(v[0]) * T(m[j][0])
        for i in range(1, 3):
            tmp = T(tmp + T(T(v[i]) * T(m[j][i])))
        out.append(T(tmp))
    return out


def mat3_det(m, T):
    """:339-343"""
    a = T(m[0][0]) * T(T(T(m[1][1]) * T(m[2][2])) - T(T(m[1][2]) * T(m[2][1])))
    b = T(m[1][0]) * T(T(T(m[0][1]) * T(m[2][2])) - T(T(m[0][2]) * T(m[2][1])))
    c = T(m[2][0]) * T(T(T(m[0][1]) * T(m[1][2])) - T(T(m[0][2]) * T(m[1][1])))
    return T(T(T(a) - T(b)) + T(c))


def mat3_inv(m, T):
    """:344-352: adjugate, every element divided by the determinant."""
    (a, b, c), (d, e, f), (g, h, i) = [[T(x) for x in row] for row in m]
    adj = [[T(e * i) - T(f * h), T(c * h) - T(b * i), T(b * f) - T(c * e)],
           [T(f * g) - T(d * i), T(a * i) - T(c * g), T(c * d) - T(a * f)],
           [T(d * h) - T(e * g), T(b * g) - T(a * h), T(a * e) - T(b * d)]]
    det = mat3_det(m, T)
    return [[T(T(x) / det) for x in row] for row in adj]


def trace_of_cov(vectors, T):
    """estimate_mean_cov(set).1.trace() (:359-378, :260-265).  `/ n as f64` on a Vec3<f32> / Mat3<f32> divides by
    `(n as f64) as f32` (:290-307)."""
    n = len(vectors)
    mean = [T(x) for x in vectors[0]]
    for v in vectors[1:]:
        mean = [T(mean[k] + T(v[k])) for k in range(3)]
    dn = T(F64(n))
    mean = [T(mean[k] / dn) for k in range(3)]
    cov = None
    for v in vectors:
        d = [T(T(v[k]) - mean[k]) for k in range(3)]
        outer = [[T(d[i] * d[j]) for j in range(3)] for i in range(3)]                # transposed_matrix (:271-281)
        cov = outer if cov is None else [[T(cov[i][j] + outer[i][j]) for j in range(3)] for i in range(3)]
    dn1 = T(F64(n - 1))
    with np.errstate(divide="ignore", invalid="ignore"):
        cov = [[T(cov[i][j] / dn1) for j in range(3)] for i in range(3)]
    # Iterator::sum of the diagonal starts from zero
    tr = T(0.0)
    for i in range(3):
        tr = T(tr + cov[i][i])
    return tr


# ---------------------------------------------------------------- types.rs
class Intrinsic:
    def __init__(self, K):
        self.m = [[F32(x) for x in row] for row in np.asarray(K, dtype=np.float32).reshape(3, 3)]
        self.inv = None

    def space_to_img(self, p):
        """:424-428"""
        r = mat_vec(self.m, p, F32)
        with np.errstate(divide="ignore", invalid="ignore"):
            return F32(r[0] / r[2]), F32(r[1] / r[2])

    def img_to_space(self, xy, z):
        """:432-445"""
        if self.inv is None:
            with np.errstate(divide="ignore", invalid="ignore"):
                self.inv = mat3_inv(self.m, F32)
        r = mat_vec(self.inv, [F32(xy[0]), F32(xy[1]), F32(1.0)], F32)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            c = F32(F32(z) / r[2])
            return [F32(r[k] * c) for k in range(3)]


def average_value_in_rect(img, sub_x, sub_y, rect) -> np.float64:
    """SubImage::average_value_in_rect (:317-339): rect = (x0, y0, x1, y1) relative to the sub-image."""
    x0, y0, x1, y1 = (int(v) for v in rect)
    count = (x1 - x0) * (y1 - y0)
    if count == 0:
        return F64(0.0)
    s = int(img[sub_y + y0: sub_y + y1, sub_x + x0: sub_x + x1].sum(dtype=np.uint64))   # u64 sum of the same pixels
    return F64(F64(s) / F64(count))


# ---------------------------------------------------------------- meanshift.rs
def build_kernel(size: int, variance) -> dict:
    """FullArray3D::build_kernel (:244-252) with kernel_function (:228-232); keyed (x, y, z)."""
    half = size // 2
    k = {}
    for z in range(size):
        for y in range(size):
            for x in range(size):
                dx, dy, dz = x - half, y - half, z - half
                norm = dx * dx + dy * dy + dz * dz
                with np.errstate(divide="ignore", invalid="ignore"):
                    k[(x, y, z)] = expf(F32(F32(F32(-1.0) * F32(norm)) / F32(F32(2.0) * F32(variance))))
    return k


def meanshift(acc: dict, init, kernel: dict, size: int, iterations: int):
    """MeanShift::meanshift for SparseArray3D<u32> (Idx = i32) (:328-407); returns (pos, trace)."""
    pos = [int(init[0]), int(init[1]), int(init[2])]
    trace = [tuple(pos)]
    half = size // 2
    for _ in range(iterations):
        num = [F32(0.0), F32(0.0), F32(0.0)]
        den = F32(0.0)
        for x in range(-half, size - half):
            for y in range(-half, size - half):
                for z in range(-half, size - half):
                    ap = (wrap_i32(pos[0] + x), wrap_i32(pos[1] + y), wrap_i32(pos[2] + z))   # (the i32::min_value() checks never fire)
                    factor = acc.get(ap, 0)
                    if factor == 0:
                        continue
                    influence = kernel[(x + half, y + half, z + half)]
                    w = F32(influence * F32(factor))
                    for k in range(3):
                        num[k] = F32(num[k] + F32(F32(ap[k]) * w))
                    den = F32(den + w)
        if den == 0.0:
            break
        with np.errstate(divide="ignore", invalid="ignore"):
            pos = [as_i32(F32(num[k] / den)) for k in range(3)]
        trace.append(tuple(pos))
    return tuple(pos), trace


# ---------------------------------------------------------------- prediction.rs
def walk(forest, img, ox, oy, t):
    """One tree: HoughTreeFunctions::binarize (houghforest.rs:185-193) from the root to a leaf."""
    cur = int(forest.roots[t])
    while cur >= 0:
        nd = forest.nodes[cur]
        a1 = average_value_in_rect(img, ox, oy, nd["r1"])
        a2 = average_value_in_rect(img, ox, oy, nd["r2"])
        cur = int(nd["child_one"]) if F64(a1 - a2) > F64(nd["threshold"]) else int(nd["child_zero"])
    return ~cur


def predict(forest, model, img, K, midp_guess=None, rot_guess=None) -> dict:
    """predict_parameter_generic (:421-493) over build_hough_cube_generic (:509-753); every intermediate is returned."""
    img = np.asarray(img, dtype=np.uint16)
    h, w = img.shape
    intr = Intrinsic(K)
    sw, sh, step = int(model.subimage_width), int(model.subimage_height), int(model.stepwidth)
    T = forest.n_trees
    pos_grid = [0] * (GUESS_GRID_PARTS * GUESS_GRID_PARTS)
    rot_grid = {}                                   # FullArray3D 20^3, index z * 400 + y * 20 + x (meanshift.rs:78-88)
    left_w, left_h = sw // 2, sh // 2
    right_w, right_h = sw - left_w, sh - left_h
    mid, rot = {}, {}
    leaf_rows, flags = [], []
    y = left_h
    while y < h - right_h:
        x = left_w
        while x < w - right_w:
            z = img[y, x]
            p3 = intr.img_to_space([F32(x), F32(y)], F32(z))
            ox, oy = x - left_w, y - left_h
            leafs = None
            if average_value_in_rect(img, ox, oy, (0, 0, sw, sh)) > 0.0:
                leafs = [walk(forest, img, ox, oy, t) for t in range(T)]
            flag = 0
            if leafs is not None:
                flag = 1
                s = F64(0.0)                                             # Iterator::sum::<f64>() starts from 0.0
                for L in leafs:
                    s = F64(s + F64(forest.leaf_prob[L]))
                prob = F64(s / F64(len(leafs)))
                if prob > 0.7:
                    flag = 3
                    for L in leafs:
                        lp = F64(forest.leaf_prob[L])
                        if not lp > 0.0:
                            continue
                        offs = forest.offsets[forest.off_begin[L]:forest.off_begin[L + 1]]
                        rots = forest.rotations[forest.rot_begin[L]:forest.rot_begin[L + 1]]
                        valtoadd = (as_usize(F64(F64(1000.0) * lp)) // len(offs)) & 0xFFFFFFFF
                        if trace_of_cov(rots, F64) <= MAX_VARIANCE_ROT:
                            for rv in rots:
                                r = []
                                for k in range(3):
                                    b = wrap_i32(as_i32(F64(F64(F64(rv[k]) * F64(ROT_GRID_PARTS)) / F64(360.0))) + ROT_GRID_PARTS // 2)
                                    if b >= ROT_GRID_PARTS:                  # inBetweenMod!: one step only
                                        b -= ROT_GRID_PARTS
                                    elif b < 0:
                                        b += ROT_GRID_PARTS
                                    r.append(b)
                                rough = [(b * GUESS_GRID_PARTS) // ROT_GRID_PARTS for b in r]
                                rot[tuple(r)] = (rot.get(tuple(r), 0) + valtoadd) & 0xFFFFFFFF
                                gi = rough[2] * 400 + rough[1] * 20 + rough[0]
                                rot_grid[gi] = (rot_grid.get(gi, 0) + valtoadd) & 0xFFFFFFFF
                        if trace_of_cov(offs, F32) <= MAX_VARIANCE_OFFSET:
                            for o in offs:
                                with np.errstate(over="ignore", invalid="ignore"):
                                    np_ = [F32(p3[k] - F32(o[k])) for k in range(3)]
                                if np_[2] < 0.0:
                                    continue
                                p2 = intr.space_to_img(np_)
                                x2 = p2[0] if p2[0] > F32(0.0) else F32(0.0)        # max!(a, b) = if a > b {a} else {b}
                                x2 = x2 if x2 < F32(w - 1) else F32(w - 1)          # min!(a, b) = if a < b {a} else {b}
                                y2 = p2[1] if p2[1] > F32(0.0) else F32(0.0)
                                y2 = y2 if y2 < F32(h - 1) else F32(h - 1)
                                z3 = F32(np_[2] / F32(ZSCALEFACTOR))
                                cell = (as_i32(np_[0]), as_i32(np_[1]), as_i32(z3))
                                mid[cell] = (mid.get(cell, 0) + valtoadd) & 0xFFFFFFFF
                                gx = as_usize(x2) * GUESS_GRID_PARTS // w
                                gy = as_usize(y2) * GUESS_GRID_PARTS // h
                                pos_grid[gy * GUESS_GRID_PARTS + gx] = (pos_grid[gy * GUESS_GRID_PARTS + gx] + valtoadd) & 0xFFFFFFFF
            leaf_rows.append(leafs if leafs is not None else [-1] * T)
            flags.append(flag)
            x += step
        y += step
    # ---- initial guesses (:694-752)
    prev_max, best_idx = 0, 0
    for idx, el in enumerate(pos_grid):
        if el > prev_max:
            prev_max, best_idx = el, idx
    gw, gh = w // GUESS_GRID_PARTS, h // GUESS_GRID_PARTS
    mxg, myg = best_idx % GUESS_GRID_PARTS, best_idx // GUESS_GRID_PARTS
    cellpx = img[gh * myg: gh * myg + gh, gw * mxg: gw * mxg + gw]
    nz = cellpx[cellpx > 0]
    meanz = F32(F64(int(nz.sum(dtype=np.uint64))) / F64(nz.size)) if nz.size > 0 else F32(0.0)
    max_x = F32(F32(F32(mxg) + F32(0.5)) * F32(gw))
    max_y = F32(F32(F32(myg) + F32(0.5)) * F32(gh))
    max3d = intr.img_to_space([max_x, max_y], meanz)
    guessmid = (as_i32(max3d[0]), as_i32(max3d[1]), as_i32(max3d[2]) // ZSCALEFACTOR)
    rx = ry = rz = oldc = 0
    for zz in range(GUESS_GRID_PARTS):                                      # FullArray3DIter: x fastest, then y, then z
        for yy in range(GUESS_GRID_PARTS):
            for xx in range(GUESS_GRID_PARTS):
                c = rot_grid.get(zz * 400 + yy * 20 + xx, 0)
                if c > 0 and c > oldc:
                    rx, ry, rz, oldc = xx, yy, zz, c
    guessrot_deg = tuple(F64(F64(F64(F64(v) * F64(360.0)) + F64(180.0)) / F64(GUESS_GRID_PARTS)) for v in (rx, ry, rz))
    # ---- predict_parameter_generic (:437-492)
    if midp_guess is not None:
        guessmid = (as_i32(F32(midp_guess[0])), as_i32(F32(midp_guess[1])), as_i32(F32(midp_guess[2])) // ZSCALEFACTOR)
    if rot_guess is not None:
        guessrot_deg = tuple(F64(F64(F64(F64(g) * F64(180.0)) / F64(3.14159)) + F64(180.0)) for g in rot_guess)
    guessrot = tuple(as_i32(F64(F64(g * F64(ROT_GRID_PARTS)) / F64(360.0))) for g in guessrot_deg)
    kernel = build_kernel(20, F32(model.gaussian_sigma))
    res_mid, trace_mid = meanshift(mid, guessmid, kernel, 20, int(model.meanshift_iterations))
    res_rot, trace_rot = meanshift(rot, guessrot, kernel, 20, int(model.meanshift_iterations))
    rotation = np.array([F64(F64(F64(F64(r) - F64(F64(ROT_GRID_PARTS) / F64(2.0))) / F64(ROT_GRID_PARTS // 2)) * F64(3.14159)) for r in res_rot])
    mid_point = np.array([F32(res_mid[0]), F32(res_mid[1]), F32(wrap_i32(res_mid[2] * ZSCALEFACTOR))], dtype=np.float32)

    def cells(d):
        a = np.array(sorted((k[0], k[1], k[2], v) for k, v in d.items()), dtype=np.int64).reshape(-1, 4)
        out = a.astype(np.int32)
        out[:, 3] = a[:, 3].astype(np.uint32).view(np.int32)
        return out

    rg = np.zeros(8000, dtype=np.uint32)
    for k, v in rot_grid.items():
        rg[k] = v
    return dict(leaf_idx=np.array(leaf_rows, dtype=np.int32).reshape(-1, T), patch_flags=np.array(flags, dtype=np.uint8),
                pos_grid=np.array(pos_grid, dtype=np.uint32), rot_grid=rg, guess_mid=np.array(guessmid, dtype=np.int32),
                guess_rot=np.array(guessrot, dtype=np.int32), mid_cells=cells(mid), rot_cells=cells(rot),
                ms_trace_mid=np.array(trace_mid, dtype=np.int32), ms_trace_rot=np.array(trace_rot, dtype=np.int32),
                mid_point=mid_point, rotation=rotation)
