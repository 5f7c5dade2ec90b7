// k_forest.hip -- once per forest: per-leaf tables, compact node tables, tree tops (k_leaf_prepare, k_nodes_compact, k_top_build)
//
// One of the kernel translation units of libdepthhead_hip.so (hand-written HIP for gfx950: wave64, 160 KB LDS/CU;
// no MFMA anywhere -- there is no dense contraction on this path).  Overview of the pipeline: dh_api.hip.
#include "dh_device.h"

// ================================================================== k_leaf_prepare
// One thread per leaf.  Restates prediction.rs:590-636 (per-leaf part) and
// meancov_estimation.rs:359-378 / :260-265 in the reference's order and types.
__global__ void __launch_bounds__(256) k_leaf_prepare(DevForest f) {
    uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= f.n_leaves) return;
    double prob = f.leaf_prob[L];
    uint32_t ob = f.off_begin[L], oe = f.off_begin[L + 1];
    uint32_t rb = f.rot_begin[L], re = f.rot_begin[L + 1];
    uint32_t n_off = oe - ob, n_rot = re - rb;
    uint32_t flags = 0, v = 0;
    float omin[3] = {INFINITY, INFINITY, INFINITY}, omax[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t bmin[3] = {255, 255, 255}, bmax[3] = {0, 0, 0};
    if (prob > 0.0 && n_off > 0 && n_rot > 0) {
        flags |= LF_PROB;
        v = (uint32_t)(f64_as_usize(__dmul_rn(1000.0, prob)) / (uint64_t)n_off);   // :594-595

        // ---- rotations, f64 (:600)
        {
            const double *s = f.rotations + (size_t)rb * 3;
            double mu[3] = {s[0], s[1], s[2]};
            for (uint32_t i = 1; i < n_rot; ++i)
                for (int k = 0; k < 3; ++k) mu[k] = __dadd_rn(mu[k], s[i * 3 + k]);
            double dn = (double)n_rot;
            for (int k = 0; k < 3; ++k) mu[k] = __ddiv_rn(mu[k], dn);
            double c0 = 0, c1 = 0, c2 = 0;  // only the diagonal feeds the trace
            for (uint32_t i = 0; i < n_rot; ++i) {
                double d0 = __dsub_rn(s[i * 3 + 0], mu[0]), d1 = __dsub_rn(s[i * 3 + 1], mu[1]),
                       d2 = __dsub_rn(s[i * 3 + 2], mu[2]);
                double q0 = __dmul_rn(d0, d0), q1 = __dmul_rn(d1, d1), q2 = __dmul_rn(d2, d2);
                if (i == 0) { c0 = q0; c1 = q1; c2 = q2; }
                else { c0 = __dadd_rn(c0, q0); c1 = __dadd_rn(c1, q1); c2 = __dadd_rn(c2, q2); }
            }
            double dn1 = (double)(n_rot - 1);
            c0 = __ddiv_rn(c0, dn1); c1 = __ddiv_rn(c1, dn1); c2 = __ddiv_rn(c2, dn1);
            double tr = __dadd_rn(__dadd_rn(__dadd_rn(0.0, c0), c1), c2);
            if (tr <= DH_MAX_VARIANCE_ROT) flags |= LF_ROT;
        }
        // ---- offsets, f32; `/ n as f64` divides by (n as f64) as f32 (meancov_estimation.rs:290-297)
        {
            const float *s = f.offsets + (size_t)ob * 3;
            float mu[3] = {s[0], s[1], s[2]};
            for (uint32_t i = 1; i < n_off; ++i)
                for (int k = 0; k < 3; ++k) mu[k] = __fadd_rn(mu[k], s[i * 3 + k]);
            float dn = (float)(double)n_off;
            for (int k = 0; k < 3; ++k) mu[k] = __fdiv_rn(mu[k], dn);
            float c0 = 0, c1 = 0, c2 = 0;
            for (uint32_t i = 0; i < n_off; ++i) {
                float d0 = __fsub_rn(s[i * 3 + 0], mu[0]), d1 = __fsub_rn(s[i * 3 + 1], mu[1]),
                      d2 = __fsub_rn(s[i * 3 + 2], mu[2]);
                float q0 = __fmul_rn(d0, d0), q1 = __fmul_rn(d1, d1), q2 = __fmul_rn(d2, d2);
                if (i == 0) { c0 = q0; c1 = q1; c2 = q2; }
                else { c0 = __fadd_rn(c0, q0); c1 = __fadd_rn(c1, q1); c2 = __fadd_rn(c2, q2); }
            }
            float dn1 = (float)(double)(n_off - 1);
            c0 = __fdiv_rn(c0, dn1); c1 = __fdiv_rn(c1, dn1); c2 = __fdiv_rn(c2, dn1);
            float tr = __fadd_rn(__fadd_rn(__fadd_rn(0.0f, c0), c1), c2);
            if (tr <= DH_MAX_VARIANCE_OFFSET) flags |= LF_OFF;
            bool finite = true;
            for (uint32_t i = 0; i < n_off; ++i)
                for (int k = 0; k < 3; ++k) {
                    float o = s[i * 3 + k];
                    if (!(fabsf(o) <= 3.0e38f)) finite = false;
                    omin[k] = fminf(omin[k], o);
                    omax[k] = fmaxf(omax[k], o);
                }
            if (!finite)
                for (int k = 0; k < 3; ++k) { omin[k] = -INFINITY; omax[k] = INFINITY; }
            bool small = finite;
            for (int k = 0; k < 3; ++k) small = small && omin[k] > -1.0e30f && omax[k] < 1.0e30f;
            if (small) flags |= LF_FIN;
        }
    }
    // ---- rotation bins (:605-632); host validation guarantees [0,120) for leaves that can vote.
    // The leaf's votes are reduced to its distinct fine bins / guess-grid cells with multiplicities
    // (v * mult wraps exactly like mult separate u32 adds of v).
    uint32_t n_fine = 0, n_rough = 0;
    for (uint32_t i = rb; i < re; ++i) {
        uint32_t packed = 0, rough = 0, mul = 1;
        for (int k = 0; k < 3; ++k) {
            int32_t r = f64_as_i32(__ddiv_rn(__dmul_rn(f.rotations[(size_t)i * 3 + k], 120.0), 360.0)) + 60;
            if (r >= DH_ROTPARTS) r -= DH_ROTPARTS;
            else if (r < 0) r += DH_ROTPARTS;
            uint32_t ru = (uint32_t)r;
            uint32_t rg = ru * DH_GRID / DH_ROTPARTS;
            uint32_t rc = ru > 255u ? 255u : ru;    // only reachable for leaves that never vote
            packed |= rc << (8 * k);
            rough += (rg < DH_GRID ? rg : 0u) * mul;   // x + 20*y + 400*z (meanshift.rs:78-88)
            mul *= DH_GRID;
            if (rc < bmin[k]) bmin[k] = rc;
            if (rc > bmax[k]) bmax[k] = rc;
        }
        uint32_t j = 0;
        for (; j < n_fine; ++j) if (f.rot_bin[rb + j] == packed) break;
        if (j == n_fine) { f.rot_bin[rb + j] = packed; f.rot_mult[rb + j] = 0; n_fine++; }
        f.rot_mult[rb + j]++;
        for (j = 0; j < n_rough; ++j) if (f.rot_rough[rb + j] == (uint16_t)rough) break;
        if (j == n_rough) { f.rot_rough[rb + j] = (uint16_t)rough; f.rough_mult[rb + j] = 0; n_rough++; }
        f.rough_mult[rb + j]++;
    }
    // The multiplicity moves into the free top byte of its bin (one load per cell in the mean shift's gathers); a bin with more
    // than 255 votes becomes several entries.  Expanded in place from the back: the entries never outnumber the leaf's votes.
    {
        uint32_t e = 0;
        for (uint32_t j = 0; j < n_fine; ++j) e += ((uint32_t)f.rot_mult[rb + j] + 254u) / 255u;
        uint32_t pos = e;
        for (uint32_t j = n_fine; j-- > 0;) {
            const uint32_t bin = f.rot_bin[rb + j] & 0xFFFFFFu;
            uint32_t m = f.rot_mult[rb + j];
            while (m) { const uint32_t part = m > 255u ? 255u : m; f.rot_bin[rb + --pos] = bin | (part << 24); m -= part; }
        }
        n_fine = e;
    }
    for (uint32_t j = 0; j < n_rough; ++j) f.rough_cell[rb + j] = (uint32_t)f.rot_rough[rb + j] | ((uint32_t)f.rough_mult[rb + j] << 16);
    f.leaf_v[L] = v;
    f.leaf_flags[L] = (uint8_t)flags;
    for (int k = 0; k < 3; ++k) { f.off_min[L * 3 + k] = omin[k]; f.off_max[L * 3 + k] = omax[k]; }
    f.rbin_box[L] = bmin[0] | (bmin[1] << 8) | (bmin[2] << 16);
    f.rbin_box_hi[L] = bmax[0] | (bmax[1] << 8) | (bmax[2] << 16);
    LeafTpl t;
    for (int k = 0; k < 3; ++k) { t.omin[k] = omin[k]; t.omax[k] = omax[k]; }
    t.v = v; t.fc = flags | (n_off << 8); t.ob = f.off4_begin[L];     // hit records index the padded 16-byte votes
    t.rlo = (flags & LF_ROT) ? f.rbin_box[L] : 0xFFFFFFFFu; t.rhi = f.rbin_box_hi[L];
    t.rb = rb; t.n_rot = n_fine | (n_rough << 16); t.flags = t.fc & 0xffu; t.prob = prob;
    f.tpl[L] = t;
    f.rot_dir[L] = make_uint4(t.rlo, t.rhi, rb, n_fine);
}

hipError_t dh_launch_leaf_prepare(const DevForest &f, hipStream_t s) {
    if (f.n_leaves == 0) return hipSuccess;
    hipLaunchKernelGGL(k_leaf_prepare, dim3((f.n_leaves + 255) / 256), dim3(256), 0, s, f);
    return hipGetLastError();
}

// ================================================================== k_nodes_compact
// Geometry-specific 16-byte node table for the uniform-rectangle fast path of k_traverse.
//
// When every split rectangle of the forest has the same size rw x rh (the in-tree trainer's
// geometry: one scale factor, hough_tree_trainer.rs:165 / prediction.rs:82-86), c1 == c2 == c and
//     avg1 - avg2 > thr   with   avg_i = fl(s_i / c),  d = fl(avg1 - avg2)
// can be decided from the INTEGER D = s1 - s2:  delta = D / c is the real difference and
// |d - delta| < 3 * 2^16 * 2^-53 < 2^-35 (each quotient is < 2^16 with relative error <= 2^-53, the
// subtraction adds one more).  So  delta >= thr + 2^-34  =>  d > thr  and  delta <= thr - 2^-34
// =>  d <= thr.  Per node: ilo = floor((thr - 2^-34) c - 2^-20), ihi = ceil((thr + 2^-34) c + 2^-20)
// (the 2^-20 pad covers the rounding of these two products); D <= ilo -> Binar::Zero, D >= ihi ->
// Binar::One, and the at most `amb` integers in between take the exact f64 path.

// Walk table (out_a, optional): the forest's nodes for walk_absorb.  Layout in 16-byte entries:
//   [0, n)              the internal nodes: {byte offsets of the two box sums (half words), ilo, Zero child, One child}
//   [n, n + n_amb)      second halves of the AMBIGUOUS nodes (below)
//   [NB = n + n_amb]    the entry every finished walk re-reads (offsets 0, never greater)
// Children are byte offsets into the table; leaf l is the virtual offset of entry NB + l, and "ambiguous at the j-th
// ambiguous node" is the virtual offset of entry NB + n_leaves + j -- anything from entry NB on ends the lock-step loop.
// A node whose threshold leaves an ambiguity band (ilo, ilo + amb] (amb_list[j] = its index; about two in a million
// real-valued thresholds, so any forest of a million nodes has some) becomes two entries: the node itself sends D <= ilo to its
// Zero child and everything else to entry n + j, which sends D > ilo + amb to the One child and the band to the ambiguity
// code; walk_absorb then decides that one visit with the reference's f64 arithmetic and walks on.
// any_amb (optional): counts the ambiguous nodes and lists the first DH_AMB_CAP of them in amb_list (the probe pass).
__global__ void __launch_bounds__(256) k_nodes_compact(const dh_node *nodes, uint32_t n, uint32_t n_leaves, int ss, int swz_log2, int swz_q, uint32_t area, NodeU *out,
                                                       NodeU *out_a, uint32_t *any_amb, uint32_t *amb_list, uint32_t n_amb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t NB = n + n_amb;
    if (i >= n) {
        if (out_a && i == n) {
            NodeU o;
            o.offs = 0; o.ilo = INT32_MAX; o.child_zero = o.child_one = (int32_t)(NB << 4);
            out_a[NB] = o;
        }
        return;
    }
    dh_node nd = nodes[i];
    const double c = (double)area, thr = nd.threshold;
    int32_t ilo;
    uint32_t amb = 0;
    if (thr >= 65535.0) {              // avg1 - avg2 <= 65535: never greater
        ilo = INT32_MAX - 16;
    } else if (thr < -65535.0) {       // avg1 - avg2 >= -65535: always greater
        ilo = INT32_MIN;
    } else {
        const double m = 5.820766091346741e-11;   // 2^-34
        const double pad = 9.5367431640625e-07;    // 2^-20
        double lo = floor(__dsub_rn(__dmul_rn(__dsub_rn(thr, m), c), pad));
        double hi = ceil(__dadd_rn(__dmul_rn(__dadd_rn(thr, m), c), pad));
        long long l = (long long)lo, h = (long long)hi;      // |thr * c| <= 65535 * 32768 < 2^31 - 2^15
        ilo = (int32_t)l;
        amb = (uint32_t)(h - l - 1);                            // 0..2
    }
    NodeU o;
    const uint32_t mm = (1u << swz_log2) - 1u;       // region slot of the rectangle's top-left cell (dh_traverse_swizzle)
    uint32_t o1 = (uint32_t)nd.r1[1] * (uint32_t)ss + (nd.r1[0] & mm) * (uint32_t)swz_q + (nd.r1[0] >> swz_log2);
    uint32_t o2 = (uint32_t)nd.r2[1] * (uint32_t)ss + (nd.r2[0] & mm) * (uint32_t)swz_q + (nd.r2[0] >> swz_log2);
    o.offs = o1 | (o2 << 14) | (amb << 28);
    o.ilo = ilo;
    o.child_zero = nd.child_zero;
    o.child_one = nd.child_one;
    if (out) out[i] = o;
    if (amb && any_amb) {
        const uint32_t k = atomicAdd(any_amb, 1u);
        if (k < DH_AMB_CAP && amb_list) amb_list[k] = i;
    }
    if (out_a) {
        o.offs = (o1 << 2) | (o2 << 18);
        const uint32_t cz = (nd.child_zero >= 0 ? (uint32_t)nd.child_zero : NB + (uint32_t)~nd.child_zero) << 4;
        const uint32_t co = (nd.child_one >= 0 ? (uint32_t)nd.child_one : NB + (uint32_t)~nd.child_one) << 4;
        o.child_zero = (int32_t)cz;
        o.child_one = (int32_t)co;
        if (amb) {
            uint32_t j = 0;
            while (j < n_amb && amb_list[j] != i) ++j;          // (rare: the few ambiguous nodes search the short list)
            NodeU o2e = o;
            o.child_one = (int32_t)((n + j) << 4);
            o2e.ilo = ilo + (int32_t)amb;
            o2e.child_zero = (int32_t)((NB + n_leaves + j) << 4);
            out_a[n + j] = o2e;
        }
        out_a[i] = o;
    }
}

hipError_t dh_launch_nodes_compact(const DevForest &f, int ss, int swz_log2, int swz_q, uint32_t area, void *out, void *out_a, uint32_t *any_amb,
                                   uint32_t *amb_list, uint32_t n_amb, hipStream_t s) {
    if (f.n_nodes == 0) return hipSuccess;
    const uint32_t n = f.n_nodes + (out_a ? 1 : 0);
    hipLaunchKernelGGL(k_nodes_compact, dim3((n + 255) / 256), dim3(256), 0, s, f.nodes, f.n_nodes, f.n_leaves, ss, swz_log2, swz_q, area, (NodeU *)out,
                       (NodeU *)out_a, any_amb, amb_list, n_amb);
    return hipGetLastError();
}

// Tree tops for walk_absorb: the first DT levels of tree t as an implicit binary heap (slot h has its Zero child at 2h + 1 and
// its One child at 2h + 2), each slot = {box-sum byte offsets, ilo} of the node there -- or an absorbing {0, INT32_MAX} when the
// path to the slot has already ended in a leaf -- followed by the walk-table entry (byte offset into nodes_a) a walk stands at after
// those DT levels.  Layout: [T][2^DT] uint2 heap (the last slot of a tree unused), then [T][2^DT] uint32 entries.
__global__ void __launch_bounds__(64) k_top_build(const NodeU *tab, const int32_t *roots, uint32_t n_nodes, uint32_t n_amb, uint32_t T, int DT, uint32_t *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t hs = 1u << DT, second = n_nodes << 4, lb = (n_nodes + n_amb) << 4;
    uint2 *heap = (uint2 *)out + (size_t)t * hs;
    uint32_t *entry = out + (size_t)T * hs * 2 + (size_t)t * hs;
    const int32_t r = roots[t];
    const uint32_t root = r >= 0 ? (uint32_t)r << 4 : lb + ((uint32_t)~r << 4);
    for (uint32_t h = 0; h < 2 * hs - 1; ++h) {
        // the bits of h + 1 below its leading one spell the path from the root: 0 = Zero child, 1 = One child.  The path stops
        // at a leaf -- and at an ambiguous node (its One child is a second-half entry), which the heap cannot hold: the walk
        // leaves the tree tops there and takes that node from the table
        uint32_t cur = root;
        bool stop = false;
        const int len = 31 - __clz((int)(h + 1));
        for (int b = len; b >= 0 && cur < lb; --b) {
            const NodeU nd = tab[cur >> 4];
            const uint32_t one = (uint32_t)nd.child_one;
            if (one >= second && one < lb) { stop = true; break; }
            if (b == 0) break;
            cur = (((h + 1) >> (b - 1)) & 1u) ? one : (uint32_t)nd.child_zero;
        }
        if (h < hs - 1) {
            const NodeU nd = tab[(stop || cur >= lb ? lb : cur) >> 4];   // (entry NB reads {0, INT32_MAX}: the path has ended)
            heap[h] = make_uint2(nd.offs, (uint32_t)nd.ilo);
        } else {
            entry[h - (hs - 1)] = cur;
        }
    }
    heap[hs - 1] = make_uint2(0u, (uint32_t)INT32_MAX);
}

hipError_t dh_launch_top_build(const DevForest &f, const void *nodes_a, uint32_t n_amb, int top_levels, uint32_t *out, hipStream_t s) {
    if (f.n_trees == 0) return hipSuccess;
    hipLaunchKernelGGL(k_top_build, dim3((f.n_trees + 63) / 64), dim3(64), 0, s, (const NodeU *)nodes_a, f.roots, f.n_nodes, n_amb, f.n_trees, top_levels, out);
    return hipGetLastError();
}
