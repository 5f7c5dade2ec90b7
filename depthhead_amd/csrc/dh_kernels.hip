// dh_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4: wave64, 160 KB LDS/CU).
//
// The hot path of depthhead's HoughPrediction::predict_parameter_parallel
// (/root/reference src/hough/prediction.rs:397-753):
//
//   k_leaf_prepare  once per forest: everything that depends only on a leaf (vote weight, both
//                   covariance gates, rotation bins, vote bounding boxes); k_nodes_compact: integer split bounds
//                   and the walk table of the uniform path; k_top_build: its tree tops for LDS
//   k_boxsum        per batch (uniform-rectangle forests): image of all rectangle sums of a frame, tile flags
//   k_traverse      per batch: tile of window positions -> region of the rectangle-sum image (or a summed-area
//                   table) in LDS -> background gate -> root->leaf walk of every (window, tree) -> window list
//   k_emit          per batch: probability gate in tree order -> (window, leaf) hit records, leaf histogram
//   k_vote          per batch: coarse 20x20 / 20^3 guess grids from the hit records
//   k_region        small batches with many hit records: first mean-shift regions gathered by several workgroups
//   k_cluster       per batch: initial guesses + both fixed-iteration Gaussian mean shifts -> pose
//   k_mask, k_hough2d, k_blur_u16, k_argmax2d: the sibling consumers (predict_mask, 2-D Hough variant);
//   k_rle_decode: BIWI run-length coded depth payloads -> frames
//
// No MFMA anywhere: there is no dense contraction on this path.  All integer work is exact and
// order-free (u32 wrapping adds); every floating-point expression is evaluated in the reference's
// type and order with explicit round-to-nearest intrinsics (no FMA contraction), so results are
// bit-identical to the CPU restatement in oracle/.  (One approximation exists, k_vote's v_rcp_f32 quotient: it only picks
// a cell of the 20 x 20 guess grid when the cell cannot depend on the rounding, and the IEEE divisions decide otherwise.)
#include "dh_internal.h"

#include <algorithm>

#define WAVE 64

// Per-phase profiling switches (kernels cut short after phase N: results INVALID) exist only in builds with
// -DDH_PROFILING_KNOBS (tools/pmc_phases.sh); the product library compiles them out.
#ifdef DH_PROFILING_KNOBS
#define KNOB_STOP(cond) (cond)
#else
#define KNOB_STOP(cond) false
#endif

// ------------------------------------------------------------------ Rust `as` casts
// float -> int truncates toward zero, saturates, NaN -> 0.
__device__ __forceinline__ int32_t f32_as_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
__device__ __forceinline__ int32_t f64_as_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483648.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return (int32_t)v;
}
__device__ __forceinline__ uint64_t f64_as_usize(double v) {
    if (v != v || v <= 0.0) return 0;
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)v;
}
__device__ __forceinline__ uint64_t f32_as_usize(float v) {
    if (v != v || v <= 0.0f) return 0;
    if (v >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)v;
}

// Mat3<f32> * Vec3<f32>: tmp = v0*m[j][0]; tmp = tmp + v_i*m[j][i]   (meancov_estimation.rs:201-216)
__device__ __forceinline__ void matvec3(const float *m, float v0, float v1, float v2, float r[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float t = __fmul_rn(v0, m[j * 3 + 0]);
        t = __fadd_rn(t, __fmul_rn(v1, m[j * 3 + 1]));
        t = __fadd_rn(t, __fmul_rn(v2, m[j * 3 + 2]));
        r[j] = t;
    }
}
// IntrinsicMatrix::img_to_space_coord (types.rs:432-445)
__device__ __forceinline__ void to3d(const float *kinv, float px, float py, float z, float out[3]) {
    float r[3];
    matvec3(kinv, px, py, 1.0f, r);
    float c = __fdiv_rn(z, r[2]);
    out[0] = __fmul_rn(r[0], c);
    out[1] = __fmul_rn(r[1], c);
    out[2] = __fmul_rn(r[2], c);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

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
struct __attribute__((aligned(16))) NodeU {
    uint32_t offs;      // LDS offset of r1's box sum | r2's << 14 | amb << 28, relative to the patch origin
    int32_t  ilo;
    int32_t  child_zero;
    int32_t  child_one;
};

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

// ================================================================== k_tile_list
// Which (frame, tile) pairs does k_traverse have to work on?  One list per x = frame mod 8 (the grid's x, which picks the XCD),
// in the order the grid visits them (frame group, then tile), entry = group << 16 | tile.  k_traverse's workgroup (x, k) takes the
// k-th entry; the workgroups beyond a list's end -- the flagged-empty tiles, four in seven on the bench frames -- then sit at the
// END of the grid, where they leave at once instead of each holding a workgroup slot (79 KB of LDS) for the 1-2 us it takes a
// workgroup to start, read its flag and go, in the middle of the real work.
__global__ void __launch_bounds__(1024) k_tile_list(const uint8_t *flags, int n_frames, int tiles, uint32_t *list, uint32_t *count, uint32_t stride) {
    __shared__ uint32_t wsum[16];
    const int x = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    const uint32_t fb = (uint32_t)(n_frames + 7) / 8, total = fb * (uint32_t)tiles;
    uint32_t base = 0;
    for (uint32_t e0 = 0; e0 < total; e0 += 1024) {
        const uint32_t e = e0 + (uint32_t)tid;
        const uint32_t z = e / (uint32_t)tiles, t = e - z * (uint32_t)tiles;
        const uint32_t frame = z * 8 + (uint32_t)x;
        const bool on = e < total && frame < (uint32_t)n_frames && flags[(size_t)frame * tiles + t] != 0;
        const unsigned long long bal = __ballot(on);
        if (lane == 0) wsum[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = 0, tot = 0;
        for (int i = 0; i < 16; ++i) { const uint32_t c = wsum[i]; off += i < wv ? c : 0u; tot += c; }
        if (on) list[(size_t)x * stride + base + off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (z << 16) | t;
        base += tot;
        __syncthreads();
    }
    if (tid == 0) count[x] = base;
}

hipError_t dh_launch_tile_list(const uint8_t *flags, int n_frames, int tiles, uint32_t *list, uint32_t *count, uint32_t stride, hipStream_t s) {
    if (n_frames == 0 || tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tile_list, dim3(8), dim3(1024), 0, s, flags, n_frames, tiles, list, count, stride);
    return hipGetLastError();
}

// ================================================================== k_traverse
// One 1024-thread workgroup per tile of PX x PY sliding-window positions of one frame: build the
// tile's image in LDS, gate out background windows, walk every tree for the active windows, write
// the leaves to the frame's window list (k_emit turns them into hit records).
//
// LDS: [ region / SAT | active windows npt u32 | their grid positions npt u32 | misc ]
//
// UNI = true (every split rectangle has one size): the image is the tile's region of the frame's
// rectangle-sum image (k_boxsum), copied straight into LDS; a node costs 2 LDS reads and an integer
// compare.  UNI = false: a summed-area table of the tile's footprint modulo 2^32 is built here (any
// rectangle inside a patch sums to < sw*sh*65535 < 2^32, checked at predictor creation, so the
// differences are exact) and one rectangle mean costs 4 LDS reads instead of the reference's O(area)
// pixel loop (types.rs:317-339).
#define TRAV_THREADS 1024
#define TRAV_WAVES (TRAV_THREADS / WAVE)
#define ROWS_IN_FLIGHT 8

// LDS image one tile works on.
// General path: the (fw + 1)-column SAT with an odd row stride (row-per-lane passes are
// bank-conflict free).
// Uniform path (rw > 0): the box-sum region of bw = fw - rw + 1 columns.  All lanes of a wave sit at
// the same tree node most of the time, so they read region cells that differ only by their
// windows' origins: `step` columns apart along a row of windows, step * ss apart between rows.
// With step = 4 a linear layout would use 16 of the 64 LDS banks.  The columns are therefore
// de-interleaved by m = the largest power of two dividing step (at most 8): cell (y, x) lives at
// y * ss + (x mod m) * q + x / m with q = ceil(bw / m) rounded up to a multiple of 4.  Window origins
// are multiples of m, so the slot of (origin + rectangle offset) is still base(origin) + offset(rectangle), neighbouring
// windows are step / m (odd) slots apart, and the row stride ss >= m * q is padded so that the
// rows of windows a wave spans land on different banks.
void dh_traverse_swizzle(int px, int step, int sw, int rw, int *swz_log2, int *swz_q, int *ss_row) {
    const int bw = (px - 1) * step + sw - rw + 1;
    int lg = 0;
    while (lg < 3 && (step & (1 << lg)) == 0) ++lg;
    const int m = 1 << lg, q = ((bw + m - 1) / m + 3) & ~3;     // planes start on 16-byte boundaries
    int best_pad = 0;
    long best_cost = -1;
    for (int pad = 0; pad < 32; pad += 4) {                         // rows too
        // bank histogram of the 64 lanes of a wave reading the same rectangle for consecutive windows
        const int ss = m * q + pad;
        int cnt[64] = {0};
        for (int i = 0; i < 64; ++i) cnt[((i / px) * step * ss + (i % px) * (step / m)) & 63]++;
        long cost = 0;
        for (int b = 0; b < 64; ++b) cost += (long)cnt[b] * cnt[b];
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_pad = pad; }
    }
    *swz_log2 = lg; *swz_q = q; *ss_row = m * q + best_pad;
}
int dh_traverse_row_stride(int px, int step, int sw, int rw) {
    if (rw > 0) { int lg, q, ss; dh_traverse_swizzle(px, step, sw, rw, &lg, &q, &ss); return ss; }
    return ((px - 1) * step + sw + 1) | 1;
}
size_t dh_traverse_lds_bytes(int px, int py, int step, int sw, int sh, int top_words, int rw, int rh) {
    size_t fh = (size_t)(py - 1) * step + sh;
    size_t ss = (size_t)dh_traverse_row_stride(px, step, sw, rw);
    size_t rows = rw > 0 ? fh - rh + 1 : fh + 1;
    size_t npt = (size_t)px * py;
    return (ss * rows + npt * 2 + 16 + (size_t)top_words) * 4;   // keep in step with the carve-up in k_traverse
}

// n / d for 0 <= n < 2^22 and d >= 1, given rd = 1.0f / d: the float estimate is off by at most one
// (relative error < 2^-22), which one correction step repairs.  Replaces the ~35-instruction integer
// division sequence in per-lane index arithmetic.
__device__ __forceinline__ int div_small(int n, int d, float rd) {
    int q = (int)((float)n * rd);
    const int r = n - q * d;
    q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    return q;
}

// Inclusive prefix sum across the 64 lanes of a wave in six DPP adds (no LDS, no barrier):
// Kogge-Stone inside each 16-lane row, then lane 15 of rows 0/2 into rows 1/3, then lane 31 into
// the upper half.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Summed-area table of a footprint of at most 64*PPL columns, built in registers: wave w owns the
// strip of R = ceil(fh/16) rows starting at w*R.  It loads its rows whole (PPL pixels per lane, fully
// coalesced), scans each row across the wave with DPP, accumulates down the strip in registers and
// publishes the strip's bottom row; after one barrier every lane adds the bottom rows of the strips
// above it and the finished rows are written once.  Three barriers, no serial stitch loop.
// Returns false when every pixel of the footprint is zero (nothing is written then).
template <int PPL>
__device__ __forceinline__ bool sat_rows_dpp(uint32_t *sat, uint32_t *flag, const uint16_t *img, int w, int fx0, int fy0,
                                             int fw, int fh, int ss) {
    constexpr int RMAX = 16 / PPL;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const int R = (fh + TRAV_WAVES - 1) / TRAV_WAVES;
    const int y0 = wave * R, x0 = lane * PPL;
    const bool al = ((w % PPL) == 0) && ((fx0 % PPL) == 0) && ((((size_t)img) & (PPL * 2 - 1)) == 0);
    uint32_t p[RMAX][PPL];
    uint32_t any_px = 0;
    // Every load is issued unconditionally (lanes outside the footprint read its first pixels and are
    // masked afterwards) so that all RMAX rows are in flight before the first one is consumed.
    const uint16_t *org = img + (size_t)fy0 * w + fx0;
    if (al) {
        const bool fullx = x0 + PPL <= fw;
        uint32_t raw[RMAX][PPL / 2];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int y = y0 + r;
            const bool ok = r < R && y < fh && fullx;
            const uint16_t *row = org + (ok ? (size_t)y * w + x0 : (size_t)0);
            if (PPL == 2) raw[r][0] = *(const uint32_t *)row;
            else { const uint2 q = *(const uint2 *)row; raw[r][0] = q.x; raw[r][PPL / 2 - 1] = q.y; }
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const bool ok = r < R && y0 + r < fh && fullx;
#pragma unroll
            for (int c = 0; c < PPL; ++c) {
                const uint32_t q = raw[r][c >> 1];
                p[r][c] = ok ? ((c & 1) ? (q >> 16) : (q & 0xffffu)) : 0u;
            }
        }
        if (x0 < fw && !fullx) {                                   // the one ragged lane at the right edge
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if (r < R && y0 + r < fh)
                    for (int c = 0; c < PPL; ++c) if (x0 + c < fw) p[r][c] = org[(size_t)(y0 + r) * w + x0 + c];
        }
    } else {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
#pragma unroll
            for (int c = 0; c < PPL; ++c) {
                const bool ok = r < R && y0 + r < fh && x0 + c < fw;
                p[r][c] = org[ok ? (size_t)(y0 + r) * w + x0 + c : (size_t)0];
            }
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
#pragma unroll
            for (int c = 0; c < PPL; ++c)
                if (!(r < R && y0 + r < fh && x0 + c < fw)) p[r][c] = 0;
        }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
#pragma unroll
        for (int c = 1; c < PPL; ++c) p[r][c] += p[r][c - 1];
        const uint32_t tot = p[r][PPL - 1];
        any_px |= tot;
        const uint32_t base = wave_incl_scan(tot) - tot;          // sum of the lanes to the left
#pragma unroll
        for (int c = 0; c < PPL; ++c) p[r][c] += base;
        if (r > 0) {
#pragma unroll
            for (int c = 0; c < PPL; ++c) p[r][c] += p[r - 1][c];   // rows beyond the strip are zero: the last row carries the total
        }
    }
    if (__ballot(any_px != 0) != 0ull && lane == 0) *flag = 1;
    uint32_t *Tb = sat;                                            // [16 strips][64 * PPL] bottom rows, inside the unwritten SAT
#pragma unroll
    for (int c = 0; c < PPL; ++c) Tb[wave * (WAVE * PPL) + x0 + c] = p[RMAX - 1][c];
    __syncthreads();
    if (*flag == 0) return false;
    uint32_t off[PPL];
#pragma unroll
    for (int c = 0; c < PPL; ++c) off[c] = 0;
    for (int w2 = 0; w2 < wave; ++w2) {
#pragma unroll
        for (int c = 0; c < PPL; ++c) off[c] += Tb[w2 * (WAVE * PPL) + x0 + c];
    }
    __syncthreads();                                               // the bottom rows are dead: the SAT may overwrite them
    for (int i = tid; i <= fw; i += TRAV_THREADS) sat[i] = 0;                     // row 0
    for (int i = tid; i < fh; i += TRAV_THREADS) sat[(i + 1) * ss] = 0;           // column 0
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int y = y0 + r;
        if (r < R && y < fh) {
            uint32_t *dst = sat + (y + 1) * ss + x0 + 1;
#pragma unroll
            for (int c = 0; c < PPL; ++c) if (x0 + c < fw) dst[c] = p[r][c] + off[c];
        }
    }
    __syncthreads();
    return true;
}

// Pass-based summed-area table (any footprint that fits LDS): vertical running sums from global
// memory by (4 columns, row segment) units, segment stitching, horizontal prefix sums in LDS.
__device__ __forceinline__ bool sat_passes(uint32_t *sat, uint32_t *flag, const uint16_t *img, int w, int fx0, int fy0,
                                           int fw, int fh, int ss) {
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    // ---- phase 1a: vertical running sums straight from global memory.  Unit = (group of 4
    // columns, segment of rows); a thread issues the loads of up to ROWS_IN_FLIGHT rows (one 8-byte
    // load each when aligned) before it touches them: one global round trip per tile, all lanes
    // busy, no cross-lane traffic.  sat[y+1][x+1] = sum of the column above within the segment.
    for (int i = tid; i <= fw; i += TRAV_THREADS) sat[i] = 0;                     // row 0
    for (int i = tid; i < fh; i += TRAV_THREADS) sat[(i + 1) * ss] = 0;           // column 0
    const bool al8 = ((w & 3) == 0) && ((fx0 & 3) == 0) && ((((size_t)img) & 7) == 0);
    const int CG = (fw + 3) >> 2;
    const int SG = max(1, min(16, TRAV_THREADS / CG));
    const int RV = (fh + SG - 1) / SG;
    uint32_t any_px = 0;
    for (int u = tid; u < CG * SG; u += TRAV_THREADS) {
        const int sg = u / CG, x = (u - sg * CG) * 4;
        const int ya = sg * RV, yb = min(fh, ya + RV);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        for (int y0 = ya; y0 < yb; y0 += ROWS_IN_FLIGHT) {
            uint32_t p[ROWS_IN_FLIGHT][4];
#pragma unroll
            for (int g = 0; g < ROWS_IN_FLIGHT; ++g) {
                const int y = y0 + g;
                p[g][0] = p[g][1] = p[g][2] = p[g][3] = 0;
                if (y < yb) {
                    const uint16_t *row = img + (size_t)(fy0 + y) * w + fx0 + x;
                    if (al8 && x + 3 < fw) {
                        uint2 q = *(const uint2 *)row;
                        p[g][0] = q.x & 0xffffu; p[g][1] = q.x >> 16; p[g][2] = q.y & 0xffffu; p[g][3] = q.y >> 16;
                    } else {
                        p[g][0] = row[0];
                        if (x + 1 < fw) p[g][1] = row[1];
                        if (x + 2 < fw) p[g][2] = row[2];
                        if (x + 3 < fw) p[g][3] = row[3];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < ROWS_IN_FLIGHT; ++g) {
                const int y = y0 + g;
                if (y < yb) {
                    r0 += p[g][0]; r1 += p[g][1]; r2 += p[g][2]; r3 += p[g][3];
                    uint32_t *dst = sat + (y + 1) * ss + x + 1;
                    dst[0] = r0;
                    if (x + 1 < fw) dst[1] = r1;
                    if (x + 2 < fw) dst[2] = r2;
                    if (x + 3 < fw) dst[3] = r3;
                }
            }
        }
        any_px |= r0 | r1 | r2 | r3;
    }
    if (__ballot(any_px != 0) != 0ull && lane == 0) *flag = 1;
    __syncthreads();
    if (*flag == 0) return false;
    // ---- phase 1b: stitch the row segments.  First the last row of every segment is made final by
    // one thread per column (a running sum over at most 16 segment totals, loads issued up front),
    // then every other row adds the final value of the segment above it: one read per unit.
    if (SG > 1) {
        for (int x = tid; x < fw; x += TRAV_THREADS) {
            uint32_t *col = sat + 1 + x;
            uint32_t tot[16];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) tot[s2] = s2 < SG ? col[min(fh, (s2 + 1) * RV) * ss] : 0u;
            uint32_t run = tot[0];
#pragma unroll
            for (int s2 = 1; s2 < 16; ++s2)
                if (s2 < SG && s2 * RV < fh) { run += tot[s2]; col[min(fh, (s2 + 1) * RV) * ss] = run; }
        }
        __syncthreads();
        {
            const int q = TRAV_THREADS / fw, r = TRAV_THREADS - q * fw;
            int sg = tid / fw, x = tid - sg * fw;
            for (; sg < SG; sg += q, x += r) {
                if (x >= fw) { x -= fw; ++sg; if (sg >= SG) break; }
                const int ya = sg * RV, yb = min(fh, ya + RV);          // pixel rows [ya, yb) = SAT rows ya+1 .. yb
                if (sg == 0 || ya >= fh) continue;
                uint32_t *col = sat + 1 + x;
                const uint32_t o = col[ya * ss];                           // final last row of the segment above
                for (int y = ya + 1; y < yb; ++y) col[y * ss] += o;       // all rows but the (already final) last
            }
        }
        __syncthreads();
    }
    // ---- phase 1c: horizontal prefix sums inside LDS, unit = (row, segment of columns); lanes
    // hold different rows and the row stride is odd, so every access is bank-conflict free.
    // Segment-last columns are stitched like the rows above.
    {
        const int SH = max(1, min(8, TRAV_THREADS / fh));
        const int CW = (fw + SH - 1) / SH;
        const int q = TRAV_THREADS / fh, r = TRAV_THREADS - q * fh;
        {
            int sg = tid / fh, y = tid - sg * fh;
            for (; sg < SH; sg += q, y += r) {
                if (y >= fh) { y -= fh; ++sg; if (sg >= SH) break; }
                uint32_t *row = sat + (1 + y) * ss + 1;
                const int xa = sg * CW, xb = min(fw, xa + CW);
                uint32_t run = 0;
#pragma unroll 4
                for (int x = xa; x < xb; ++x) { run += row[x]; row[x] = run; }
            }
        }
        __syncthreads();
        if (SH > 1) {
            for (int y = tid; y < fh; y += TRAV_THREADS) {
                uint32_t *row = sat + (1 + y) * ss;
                uint32_t tot[8];
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) tot[s2] = s2 < SH ? row[min(fw, (s2 + 1) * CW)] : 0u;
                uint32_t run = tot[0];
#pragma unroll
                for (int s2 = 1; s2 < 8; ++s2)
                    if (s2 < SH && s2 * CW < fw) { run += tot[s2]; row[min(fw, (s2 + 1) * CW)] = run; }
            }
            __syncthreads();
            int sg = tid / fh, y = tid - sg * fh;
            for (; sg < SH; sg += q, y += r) {
                if (y >= fh) { y -= fh; ++sg; if (sg >= SH) break; }
                const int xa = sg * CW, xb = min(fw, xa + CW);           // pixel columns [xa, xb) = SAT columns xa+1 .. xb
                if (sg == 0 || xa >= fw) continue;
                uint32_t *row = sat + (1 + y) * ss;
                const uint32_t o = row[xa];
                for (int x = xa + 1; x < xb; ++x) row[x] += o;
            }
            __syncthreads();
        }
    }
    return true;
}

// ================================================================== k_pixflags
// General (mixed-rectangle) path: which k_traverse tiles have a non-zero pixel under their footprint?
// One workgroup scans a band of 32 image rows of one frame (8-byte loads, 4 columns per lane); a lane
// that saw a non-zero pixel marks the tiles whose footprints contain its columns and the band's rows
// (plain stores of 1; the flags are zeroed per batch).  Conservative by construction.
__global__ void __launch_bounds__(256) k_pixflags(PixFlagArgs a) {
    const int frame = (int)blockIdx.z * 8 + (int)blockIdx.x;
    if (frame >= a.n_frames) return;
    const int y0 = (int)blockIdx.y * 32, y1 = min(y0 + 32, a.h);
    const uint16_t *img = a.frames + (size_t)frame * a.w * a.h;
    uint8_t *flags = a.tile_flags + (size_t)frame * a.tiles_x * a.tiles_y;
    const float r_tpx = 1.0f / (float)a.tpx, r_tpy = 1.0f / (float)a.tpy;
    const bool al = (a.w & 3) == 0 && (((size_t)a.frames) & 7) == 0;
    for (int x = 4 * (int)threadIdx.x; x < a.w; x += 4 * 256) {
        uint32_t acc = 0;
        if (al) {
            for (int y = y0; y < y1; ++y) { const uint2 q = *(const uint2 *)(img + (size_t)y * a.w + x); acc |= q.x | q.y; }
        } else {
            for (int y = y0; y < y1; ++y)
                for (int c = 0; c < 4 && x + c < a.w; ++c) acc |= img[(size_t)y * a.w + x + c];
        }
        if (!acc) continue;
        // tile tx covers pixel columns [tx * tpx, tx * tpx + tfw), rows likewise
        const int tx1 = min(div_small(min(x + 3, a.w - 1), a.tpx, r_tpx), a.tiles_x - 1), tx0 = x - a.tfw < 0 ? 0 : div_small(x - a.tfw, a.tpx, r_tpx) + 1;
        const int ty1 = min(div_small(y1 - 1, a.tpy, r_tpy), a.tiles_y - 1), ty0 = y0 - a.tfh < 0 ? 0 : div_small(y0 - a.tfh, a.tpy, r_tpy) + 1;
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) flags[ty * a.tiles_x + tx] = 1;
    }
}

hipError_t dh_launch_pixflags(const PixFlagArgs &a, hipStream_t s) {
    const int fb = (a.n_frames + 7) / 8, bands = (a.h + 31) / 32;
    if (fb == 0 || bands == 0) return hipSuccess;
    if (bands > 65535 || fb > 65535) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_pixflags, dim3(8, bands, fb), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ================================================================== k_boxsum
// Uniform-rectangle forests (the trainer's geometry, types.rs:82-91): every split test compares the
// sums of two rw x rh rectangles, so the image of ALL such sums is computed once per frame here
// and k_traverse only copies the region under its tile.  out[y][x] = sum of the pixels of the
// rectangle whose top-left pixel is (x, y) -- an exact integer < 2^32, so any order of summation
// gives the reference's value (types.rs:317-339 adds the same pixels one by one).
//
// Streaming, barrier-free: one WAVE owns a band of `oh` output rows x 256 image columns (4 per
// lane) and marches down it.  Per row it adds the entering image row to / subtracts the leaving
// row from its per-column running sums V (the rh-row vertical sums), prefix-sums V across the wave
// (3 adds + the 6-DPP scan), parks the exclusive prefix in a wave-private LDS row and reads it back
// rw columns to the right: out[x] = Pex[x + rw] - Pex[x].  One 8-byte load per image row and lane,
// one 16-byte store per output row and lane.
#define BOXW_THREADS 256
#define BOXW_WAVES (BOXW_THREADS / WAVE)
#define BOX_SPAN 256             // image columns per wave
#define BOX_MAXR 96              // largest rectangle edge (host: kBoxMaxRect)
#ifndef BOX_ROWS_IN_FLIGHT
#define BOX_ROWS_IN_FLIGHT 4     // (measured r02, 256 VGA frames: 2 rows 0.099 ms, 4 rows 0.092, 8 rows 0.090)
#endif

// AL: 8-byte row loads (w % 4 == 0, 8-byte aligned frames: a lane is all inside or all outside the
// image); RW4: rw % 4 == 0 (the shifted prefix is read back with one 16-byte LDS load).  Both are
// compile-time (the host picks the instance) so that the loads of a group stay straight-line code
// and every instance gets its own register budget.
template <bool AL, bool RW4, int RIF, bool RING>
__device__ __forceinline__ void boxsum_wave(const BoxArgs &a, int frame, const uint16_t *img, uint32_t *pex, uint2 *ring, uint32_t *out,
                                            int lane, int x, int Y0, int y_end, bool store, int part) {
    // Columns right of the image read a.zeros with row stride 0 instead of being masked, so every
    // load is unconditional and nothing has to wait for it before its use.
    const bool in0 = x < a.w, in1 = x + 1 < a.w, in2 = x + 2 < a.w, in3 = x + 3 < a.w;
    const uint16_t *c0 = in0 ? img + x : a.zeros, *c1 = in1 ? img + x + 1 : a.zeros;
    const uint16_t *c2 = in2 ? img + x + 2 : a.zeros, *c3 = in3 ? img + x + 3 : a.zeros;
    const size_t s0 = in0 ? a.w : 0, s1 = in1 ? a.w : 0, s2 = in2 ? a.w : 0, s3 = in3 ? a.w : 0;
    auto load_row = [&](int y) -> uint2 {
        if (AL) return *(const uint2 *)(c0 + (size_t)y * s0);
        return make_uint2((uint32_t)c0[(size_t)y * s0] | ((uint32_t)c1[(size_t)y * s1] << 16),
                          (uint32_t)c2[(size_t)y * s2] | ((uint32_t)c3[(size_t)y * s3] << 16));
    };
    // Step t brings image row Y0 + t into the rh-row window; from t = rh - 1 on it also emits output
    // row Y0 + t - (rh - 1) and then drops that row from the window.  Steps run in groups of RIF
    // (rows in flight; 1 on the 2-byte-load path to stay within 64 VGPRs) whose loads are issued one
    // whole group ahead of their use.  The row that leaves the window was loaded rh - 1 steps
    // earlier: with RING it is kept in a wave-private LDS ring of rh - 1 packed rows (so every pixel
    // crosses the memory system once); without, it is simply fetched again (during the warm-up steps
    // those fetches read row Y0 and are ignored).
    const int nsteps = (a.rh - 1) + (y_end - Y0), warm = a.rh - 1;
    const uint32_t *pex_rd = pex + 4 * lane + a.rw;
    // output slots of this lane's four columns within a row: plane (x mod m), position x / m
    const int mm = (1 << a.lg) - 1;
    const size_t row_pitch = (size_t)a.plane << a.lg;
    const int o0 = (x & mm) * a.plane + (x >> a.lg), o1 = ((x + 1) & mm) * a.plane + ((x + 1) >> a.lg);
    const int o2 = ((x + 2) & mm) * a.plane + ((x + 2) >> a.lg), o3 = ((x + 3) & mm) * a.plane + ((x + 3) >> a.lg);
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, acc = 0;
    uint8_t *flags = a.tile_flags + (size_t)frame * a.tiles_x * a.tiles_y;
    const float r_tpx = 1.0f / (float)a.tpx, r_tpy = 1.0f / (float)a.tpy;
    uint2 e[RIF], l[RIF], en[RIF], ln[RIF];
    int rs = 0;                          // ring slot of step t: t mod (rh - 1)
    // sparse stores (BoxArgs::blk_mask): the previous fill's non-zero mask of the 32-row block being emitted, and the next block's
    unsigned long long *bm = a.blk_mask ? a.blk_mask + ((size_t)frame * a.mask_blocks) * a.parts + part : nullptr;
    unsigned long long pblk = bm ? bm[(size_t)(Y0 >> 5) * a.parts] : ~0ull;
    unsigned long long pnext = bm && ((Y0 >> 5) + 1) < a.mask_blocks ? bm[(size_t)((Y0 >> 5) + 1) * a.parts] : ~0ull;
#pragma unroll
    for (int k = 0; k < RIF; ++k) {
        const int t = min(k, nsteps - 1);
        e[k] = load_row(Y0 + t);
        l[k] = RING ? make_uint2(0u, 0u) : load_row(Y0 + max(t - warm, 0));
    }
    for (int t0 = 0; t0 < nsteps; t0 += RIF) {
#pragma unroll
        for (int k = 0; k < RIF; ++k) {
            const int t = min(t0 + RIF + k, nsteps - 1);
            en[k] = load_row(Y0 + t);
            ln[k] = RING ? make_uint2(0u, 0u) : load_row(Y0 + max(t - warm, 0));
        }
#pragma unroll
        for (int k = 0; k < RIF; ++k) {
            const int t = t0 + k;
            if (t >= nsteps) break;
            v0 += e[k].x & 0xffffu; v1 += e[k].x >> 16; v2 += e[k].y & 0xffffu; v3 += e[k].y >> 16;
            if (RING && warm > 0) {
                // swap the entering row into the slot of the row that leaves at this step
                if (t >= warm) l[k] = ring[rs * WAVE + lane];
                ring[rs * WAVE + lane] = e[k];
                if (++rs == warm) rs = 0;
            }
            if (t < warm) continue;
            const uint32_t e1 = v0, e2 = v0 + v1, e3 = e2 + v2, tot = e3 + v3;
            const uint32_t base = wave_incl_scan(tot) - tot;               // sum of the columns left of this lane
            const uint4 pe = make_uint4(base, base + e1, base + e2, base + e3);
            __builtin_amdgcn_wave_barrier();                               // the previous row's reads are issued
            *(uint4 *)(pex + 4 * lane) = pe;
            __builtin_amdgcn_wave_barrier();                               // LDS is in order within a wave
            uint4 r;
            if (RW4) r = *(const uint4 *)__builtin_assume_aligned(pex_rd, 16);
            else r = make_uint4(pex_rd[0], pex_rd[1], pex_rd[2], pex_rd[3]);
            const uint32_t nzr = (r.x - pe.x) | (r.y - pe.y) | (r.z - pe.z) | (r.w - pe.w);
            if (store) acc |= nzr;
            if (store && (nzr != 0 || ((pblk >> lane) & 1ull))) {          // (zero over zero is not written again)
                uint32_t *orow = out + (size_t)(Y0 + t - warm) * row_pitch;
                if (a.lg == 0) *(uint4 *)(orow + o0) = make_uint4(r.x - pe.x, r.y - pe.y, r.z - pe.z, r.w - pe.w);
                else if (a.lg == 1) {                      // two planes: columns x, x + 2 and x + 1, x + 3 are neighbours in theirs
                    *(uint2 *)(orow + o0) = make_uint2(r.x - pe.x, r.z - pe.z);
                    *(uint2 *)(orow + o1) = make_uint2(r.y - pe.y, r.w - pe.w);
                } else { orow[o0] = r.x - pe.x; orow[o1] = r.y - pe.y; orow[o2] = r.z - pe.z; orow[o3] = r.w - pe.w; }
            }
            v0 -= l[k].x & 0xffffu; v1 -= l[k].x >> 16; v2 -= l[k].y & 0xffffu; v3 -= l[k].y >> 16;
            // Which k_traverse tiles have a non-zero rectangle sum in their region?  Every 32 output rows
            // (and at the end of the band) each lane that saw a non-zero sum marks the tiles whose regions
            // contain its columns and those rows (plain stores of 1: the flags are zeroed per batch).
            const int yo = Y0 + t - warm;
            if ((yo & 31) == 31 || yo == y_end - 1) {
                if (bm) {
                    const unsigned long long nm = __ballot(acc != 0);
                    if (lane == 0) bm[(size_t)(yo >> 5) * a.parts] = nm;
                    pblk = pnext;
                    pnext = ((yo >> 5) + 2) < a.mask_blocks ? bm[(size_t)((yo >> 5) + 2) * a.parts] : ~0ull;
                }
                if (acc != 0) {
                    // tile tx covers columns [tx * tpx, tx * tpx + tbw): tx in [(x + 3 - tbw) / tpx + 1 .. x / tpx] clipped
                    const int y_lo = max(yo & ~31, Y0);
                    const int tx1 = min(div_small(x + 3, a.tpx, r_tpx), a.tiles_x - 1), tx0 = max(x - a.tbw < 0 ? 0 : div_small(x - a.tbw, a.tpx, r_tpx) + 1, 0);
                    const int ty1 = min(div_small(yo, a.tpy, r_tpy), a.tiles_y - 1), ty0 = max(y_lo - a.tbh + 1 <= 0 ? 0 : div_small(y_lo - a.tbh, a.tpy, r_tpy) + 1, 0);
                    for (int ty = ty0; ty <= ty1; ++ty)
                        for (int tx = tx0; tx <= tx1; ++tx) flags[ty * a.tiles_x + tx] = 1;
                }
                acc = 0;
            }
        }
#pragma unroll
        for (int k = 0; k < RIF; ++k) { e[k] = en[k]; l[k] = ln[k]; }
    }
}

template <bool AL, bool RW4, bool RING>
__global__ void __launch_bounds__(BOXW_THREADS) k_boxsum(BoxArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t pex_s[BOXW_WAVES][BOX_SPAN + BOX_MAXR + 8];
    extern __shared__ __attribute__((aligned(16))) uint32_t box_dyn[];       // RING: [BOXW_WAVES][rh - 1][64] packed rows
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const int frame = (int)blockIdx.z * 8 + (int)blockIdx.x;     // grid (8, blocks per frame, frames / 8): same frame -> XCD mapping as k_traverse
    const int unit = (int)blockIdx.y * BOXW_WAVES + wv;
    if (frame >= a.n_frames || unit >= a.bands * a.parts) return;      // waves are independent: no barriers below
    const int band = div_small(unit, a.parts, 1.0f / (float)a.parts), part = unit - band * a.parts;
    const int X0 = part * a.ow, Y0 = band * a.oh;           // X0 % 4 == 0 (host)
    const int y_end = min(Y0 + a.oh, a.rows);
    if (Y0 >= a.rows) return;
    const int x = X0 + 4 * lane;                            // this lane's columns x .. x + 3
    const uint16_t *img = a.frames + (size_t)frame * a.w * a.h;
    const bool store = 4 * lane < a.ow && x + 3 < (a.plane << a.lg);   // columns right of w - rw get clipped-rectangle sums or stay 0
    uint32_t *pex = pex_s[wv];
    uint2 *ring = (uint2 *)box_dyn + (size_t)wv * (a.rh - 1) * WAVE;
    uint32_t *out = a.out + (size_t)frame * a.rows * ((size_t)a.plane << a.lg);
    boxsum_wave<AL, RW4, AL ? BOX_ROWS_IN_FLIGHT : 1, RING>(a, frame, img, pex, ring, out, lane, x, Y0, y_end, store, part);
}

hipError_t dh_launch_boxsum(const BoxArgs &a, hipStream_t s) {
    const int fb = (a.n_frames + 7) / 8;
    if (fb == 0 || a.blocks_per_frame == 0) return hipSuccess;
    if (a.blocks_per_frame > 65535 || fb > 65535) return hipErrorInvalidConfiguration;
    // 8-byte row loads need w % 4 == 0 and 8-byte aligned frames; rw % 4 == 0 gives a 16-byte LDS read-back;
    // a.ring: the host sized the bands for the LDS-ring instance (rh - 1 packed rows per wave)
    const bool al = (a.w & 3) == 0 && (((size_t)a.frames) & 7) == 0, rw4 = (a.rw & 3) == 0;
    const dim3 grid(8, a.blocks_per_frame, fb), block(BOXW_THREADS);
    const size_t ring_bytes = a.ring ? (size_t)BOXW_WAVES * (a.rh - 1) * WAVE * sizeof(uint2) : 0;
    if (a.ring) {
        if (al && rw4) hipLaunchKernelGGL((k_boxsum<true, true, true>), grid, block, ring_bytes, s, a);
        else if (al) hipLaunchKernelGGL((k_boxsum<true, false, true>), grid, block, ring_bytes, s, a);
        else if (rw4) hipLaunchKernelGGL((k_boxsum<false, true, true>), grid, block, ring_bytes, s, a);
        else hipLaunchKernelGGL((k_boxsum<false, false, true>), grid, block, ring_bytes, s, a);
    } else {
        if (al && rw4) hipLaunchKernelGGL((k_boxsum<true, true, false>), grid, block, 0, s, a);
        else if (al) hipLaunchKernelGGL((k_boxsum<true, false, false>), grid, block, 0, s, a);
        else if (rw4) hipLaunchKernelGGL((k_boxsum<false, true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_boxsum<false, false, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

// Root-to-leaf walks of the uniform path: work item k = (tree k / n_active, active slot k % n_active);
// a lane takes items tid, tid + 1024, ... W at a time.  Leaf ids go to the tile's segment of the frame's
// window list (wleaf, read by k_emit) and, when the taps are on, to the dense [position][tree] array.  HoughTreeFunctions::binarize
// (houghforest.rs:185-193) on two rectangle sums with the integer test of NodeU, falling back to
// the reference's own f64 arithmetic inside the band the integer test cannot decide.
template <int W>
__device__ __forceinline__ void walk_uniform(const TraverseArgs &a, const uint32_t *sat, int32_t *wleaf, int32_t *dleaf, const uint32_t *active,
                                             const uint32_t *agp, int n_active, int total, int cx, int ss, int T) {
    const int tid = threadIdx.x;
    const NodeU *nodes_u = (const NodeU *)a.nodes_u;
    const float r_active = 1.0f / (float)n_active, r_cx = 1.0f / (float)cx;
    for (int k0 = tid; k0 < total; k0 += W * TRAV_THREADS) {
        int cur[W], dst[W], ddst[W];
        const uint32_t *sp[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int k = k0 + i * TRAV_THREADS;
            const bool has = k < total;
            const int kk = has ? k : k0;
            const int t = div_small(kk, n_active, r_active), slot = kk - t * n_active;
            const int p = (int)active[slot];
            const int py = div_small(p, cx, r_cx), px = p - py * cx;
            sp[i] = sat + py * a.step * ss + ((px * a.step) >> a.swz_log2);   // window origins are multiples of m
            dst[i] = t * a.win_cap + slot;                  // window list: [tree][slot], consecutive lanes -> consecutive words
            ddst[i] = dleaf ? (int)agp[slot] * T + t : 0;   // dense tap: [window position][tree]
            cur[i] = has ? a.f.roots[t] : -1;
        }
        for (;;) {
            bool go = false;
#pragma unroll
            for (int i = 0; i < W; ++i) go |= cur[i] >= 0;
            if (!go) break;
            uint4 n[W];
#pragma unroll
            for (int i = 0; i < W; ++i) n[i] = *(const uint4 *)(nodes_u + (cur[i] >= 0 ? cur[i] : 0));   // a finished walk re-reads node 0 harmlessly
            uint32_t s1[W], s2[W];
#pragma unroll
            for (int i = 0; i < W; ++i) { s1[i] = sp[i][n[i].x & 0x3fffu]; s2[i] = sp[i][(n[i].x >> 14) & 0x3fffu]; }
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const int32_t d = (int32_t)s1[i] - (int32_t)s2[i];
                bool one = d > (int32_t)n[i].y;
                const uint32_t amb = n[i].x >> 28;
                if (cur[i] >= 0 && amb && one && d <= (int32_t)n[i].y + (int32_t)amb) {
                    const double thr = a.f.nodes[cur[i]].threshold, c = (double)a.area;
                    one = __dsub_rn(__ddiv_rn((double)s1[i], c), __ddiv_rn((double)s2[i], c)) > thr;   // types.rs:338
                }
                if (cur[i] >= 0) cur[i] = one ? (int)n[i].w : (int)n[i].z;
            }
        }
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (k0 + i * TRAV_THREADS < total) {
                wleaf[dst[i]] = ~cur[i];
                if (dleaf) dleaf[ddst[i]] = ~cur[i];
            }
    }
}

// The same walks over the walk table nodes_a (k_nodes_compact's second output; used unless the forest has more than
// DH_AMB_CAP ambiguous nodes).  What bounds a level of these walks is the throughput of the 16-byte node gather, not the
// arithmetic -- cutting the loop from 42 to 17 VALU instructions alone changed nothing
// (profiles/r02_traverse_experiments.md, r02_ubench_gather_occ.txt) -- so:
//  * the first top_levels levels of every tree are walked from an LDS copy (k_top_build: implicit heap, 8-byte slots, no child
//    pointers), which takes a third of the gathers off that path;
//  * children are byte offsets into the table (the load needs no address arithmetic), leaf l is the virtual offset of entry
//    NB + l, and a finished walk re-reads entry NB, the same line for every finished lane: a gather costs by the distinct
//    lines its lanes touch;
//  * the loop is wave-uniform (no per-walk exec masks): two half-word byte offsets -> two LDS reads, subtract, compare, select;
//  * a walk that lands in the ambiguity band of a node leaves the loop with that node's code, is decided by the reference's
//    f64 arithmetic and re-enters (rare; the layout is described at k_nodes_compact).
template <int W>
__device__ __forceinline__ void walk_absorb(const TraverseArgs &a, const uint32_t *sat, const uint32_t *top, int32_t *wleaf, int32_t *dleaf,
                                            const uint32_t *active, const uint32_t *agp, int n_active, int total, int cx, int ss, int T) {
    const int tid = threadIdx.x;
    const char *tab = (const char *)a.nodes_a;
    const uint32_t lb = a.walk_lb;                   // byte offset of the entry finished walks re-read; leaf l = lb + 16 l
    const uint32_t amb_base = lb + (a.f.n_leaves << 4);   // codes "ambiguous at the j-th ambiguous node" (k_nodes_compact)
    const int DT = a.top_levels;
    const uint32_t hs = 1u << DT;
    const char *entries = (const char *)(top + (size_t)T * hs * 2);
    const float r_active = 1.0f / (float)n_active, r_cx = 1.0f / (float)cx;
    for (int k0 = tid; k0 < total; k0 += W * TRAV_THREADS) {
        uint32_t cur[W];
        int dst[W], ddst[W];
        const char *sp[W];
        uint32_t hb[W], h[W];      // byte offset of the tree's heap in the LDS copy of the tree tops; slot in it
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int k = k0 + i * TRAV_THREADS;
            const int kk = k < total ? k : k0;
            const int t = div_small(kk, n_active, r_active), slot = kk - t * n_active;
            const int p = (int)active[slot];
            const int py = div_small(p, cx, r_cx), px = p - py * cx;
            sp[i] = (const char *)(sat + py * a.step * ss + ((px * a.step) >> a.swz_log2));   // window origins are multiples of m
            dst[i] = t * a.win_cap + slot;
            ddst[i] = dleaf ? (int)agp[slot] * T + t : 0;
            hb[i] = ((uint32_t)t * hs) << 3;
            h[i] = 0;
        }
        // the tree tops: levels 0 .. DT-1 from LDS (children are implicit; a path that has ended sits on absorbing slots)
        for (int l = 0; l < DT; ++l) {
            uint2 nd[W];
#pragma unroll
            for (int i = 0; i < W; ++i) nd[i] = *(const uint2 *)((const char *)top + hb[i] + (h[i] << 3));
            uint32_t s1[W], s2[W];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                s1[i] = *(const uint32_t *)(sp[i] + (nd[i].x & 0xffffu));
                s2[i] = *(const uint32_t *)(sp[i] + (nd[i].x >> 16));
            }
#pragma unroll
            for (int i = 0; i < W; ++i) h[i] = 2 * h[i] + (((int32_t)s1[i] - (int32_t)s2[i] > (int32_t)nd[i].y) ? 2u : 1u);
        }
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const uint32_t e = *(const uint32_t *)(entries + (hb[i] >> 1) + ((h[i] - (hs - 1)) << 2));
            cur[i] = k0 + i * TRAV_THREADS < total ? e : lb;
        }
        for (;;) {
            for (;;) {
                uint32_t mn = cur[0];
#pragma unroll
                for (int i = 1; i < W; ++i) mn = min(mn, cur[i]);
                if (__ballot(mn < lb) == 0ull) break;
                uint4 n[W];
#pragma unroll
                for (int i = 0; i < W; ++i) n[i] = *(const uint4 *)(tab + min(cur[i], lb));   // finished walks all re-read ONE entry: a gather costs by its distinct lines
                uint32_t s1[W], s2[W];
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    s1[i] = *(const uint32_t *)(sp[i] + (n[i].x & 0xffffu));
                    s2[i] = *(const uint32_t *)(sp[i] + (n[i].x >> 16));
                }
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const uint32_t nx = ((int32_t)s1[i] - (int32_t)s2[i] > (int32_t)n[i].y) ? n[i].w : n[i].z;
                    cur[i] = cur[i] >= lb ? cur[i] : nx;
                }
            }
            // a walk that stands in the ambiguity band of a node (rare: see k_nodes_compact) is decided there by the reference's
            // own f64 arithmetic (types.rs:338, houghforest.rs:188-191) and walks on
            bool again = false;
#pragma unroll
            for (int i = 0; i < W; ++i)
                if (cur[i] >= amb_base) {
                    const uint32_t j = (cur[i] - amb_base) >> 4, X = a.amb_list[j];
                    const uint4 nu = *(const uint4 *)(tab + ((size_t)X << 4)), n2 = *(const uint4 *)(tab + ((size_t)(a.f.n_nodes + j) << 4));
                    const uint32_t s1 = *(const uint32_t *)(sp[i] + (nu.x & 0xffffu)), s2 = *(const uint32_t *)(sp[i] + (nu.x >> 16));
                    const double thr = a.f.nodes[X].threshold, c = (double)a.area;
                    const bool one = __dsub_rn(__ddiv_rn((double)s1, c), __ddiv_rn((double)s2, c)) > thr;
                    cur[i] = one ? n2.w : nu.z;
                    again = true;
                }
            if (__ballot(again) == 0ull) break;
        }
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (k0 + i * TRAV_THREADS < total) {
                const int32_t leaf = (int32_t)((cur[i] - lb) >> 4);
                wleaf[dst[i]] = leaf;
                if (dleaf) dleaf[ddst[i]] = leaf;
            }
    }
}

// GI (general path, patches of at most 255 x 255): split tests decided on integers.  With C_i = max(c_i, 1) (an empty
// rectangle sums to 0, types.rs:335-338) the real difference of the two means is delta = (s1 C2 - s2 C1) / (C1 C2), and the
// reference's d = fl(fl(s1 / c1) - fl(s2 / c2)) satisfies |d - delta| < 2^-35 as on the uniform path, so
// D = s1 C2 - s2 C1 (64-bit) is compared with per-node bounds ilo = floor((thr - 2^-34) C1 C2 - pad) and
// ihi = ilo + 1 + amb (NodeG, built on the host): D <= ilo -> Zero, D >= ihi -> One, anything between takes the reference's
// own two f64 divisions.  Replaces ~60 VALU instructions of f64 division per visit by two 32 x 32 -> 64 multiplies.

// Root-to-leaf walks of the general path with the integer split test of NodeG (see k_traverse), W walks per lane in lock
// step like walk_uniform.  sat is the tile's summed-area table modulo 2^32 with row stride ss.
template <int W>
__device__ __forceinline__ void walk_general_int(const TraverseArgs &a, const uint32_t *sat, int32_t *wleaf, int32_t *dleaf, const uint32_t *active,
                                                 const uint32_t *agp, int n_active, int total, int cx, int ss, int T) {
    const int tid = threadIdx.x;
    const NodeG *nodes_g = (const NodeG *)a.nodes_g;
    const float r_active = 1.0f / (float)n_active, r_cx = 1.0f / (float)cx;
    for (int k0 = tid; k0 < total; k0 += W * TRAV_THREADS) {
        int cur[W], dst[W], ddst[W];
        const uint32_t *sp[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int k = k0 + i * TRAV_THREADS;
            const bool has = k < total;
            const int kk = has ? k : k0;
            const int t = div_small(kk, n_active, r_active), slot = kk - t * n_active;
            const int p = (int)active[slot];
            const int py = div_small(p, cx, r_cx), px = p - py * cx;
            sp[i] = sat + py * a.step * ss + px * a.step;
            dst[i] = t * a.win_cap + slot;
            ddst[i] = dleaf ? (int)agp[slot] * T + t : 0;
            cur[i] = has ? a.f.roots[t] : -1;
        }
        for (;;) {
            bool go = false;
#pragma unroll
            for (int i = 0; i < W; ++i) go |= cur[i] >= 0;
            if (!go) break;
            uint4 n0[W], n1[W];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const uint4 *np = (const uint4 *)(nodes_g + (cur[i] >= 0 ? cur[i] : 0));       // a finished walk re-reads node 0 harmlessly
                n0[i] = np[0]; n1[i] = np[1];
            }
            uint32_t c[W][8];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const int ax0 = n0[i].x & 0xff, ay0 = (n0[i].x >> 8) & 0xff, ax1 = (n0[i].x >> 16) & 0xff, ay1 = n0[i].x >> 24;
                const int bx0 = n0[i].y & 0xff, by0 = (n0[i].y >> 8) & 0xff, bx1 = (n0[i].y >> 16) & 0xff, by1 = n0[i].y >> 24;
                const uint32_t *q = sp[i];
                c[i][0] = q[ay1 * ss + ax1]; c[i][1] = q[ay0 * ss + ax1]; c[i][2] = q[ay1 * ss + ax0]; c[i][3] = q[ay0 * ss + ax0];
                c[i][4] = q[by1 * ss + bx1]; c[i][5] = q[by0 * ss + bx1]; c[i][6] = q[by1 * ss + bx0]; c[i][7] = q[by0 * ss + bx0];
            }
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const uint32_t s1 = c[i][0] - c[i][1] - c[i][2] + c[i][3], s2 = c[i][4] - c[i][5] - c[i][6] + c[i][7];
                const uint32_t C1 = n1[i].w & 0xffffu, C2 = n1[i].w >> 16;
                const long long D = (long long)((unsigned long long)s1 * C2) - (long long)((unsigned long long)s2 * C1);
                const long long ilo = (long long)(((unsigned long long)n0[i].w << 32) | n0[i].z);
                bool one = D > ilo;
                if (cur[i] >= 0 && one && (unsigned long long)(D - ilo - 1) < (unsigned long long)n1[i].z) {
                    // inside the band the integers cannot decide: the reference's own arithmetic (types.rs:335-338, houghforest.rs:188-191)
                    const dh_node nd = a.f.nodes[cur[i]];
                    const uint32_t c1 = (uint32_t)((nd.r1[2] - nd.r1[0]) * (nd.r1[3] - nd.r1[1])), c2 = (uint32_t)((nd.r2[2] - nd.r2[0]) * (nd.r2[3] - nd.r2[1]));
                    const double a1 = c1 ? __ddiv_rn((double)s1, (double)c1) : 0.0;
                    const double a2 = c2 ? __ddiv_rn((double)s2, (double)c2) : 0.0;
                    one = __dsub_rn(a1, a2) > nd.threshold;
                }
                if (cur[i] >= 0) cur[i] = one ? (int)n1[i].y : (int)n1[i].x;
            }
        }
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (k0 + i * TRAV_THREADS < total) {
                wleaf[dst[i]] = ~cur[i];
                if (dleaf) dleaf[ddst[i]] = ~cur[i];
            }
    }
}

#ifdef DH_PROFILING_KNOBS
#define STAMP(k)                                                                        \
    if (a.dbg_stamps && tid == 0) {                                                     \
        unsigned long long t_ = clock64();                                              \
        atomicAdd(&a.dbg_stamps[k], t_ - t_prev);                                        \
        t_prev = t_;                                                                    \
    }
#else
#define STAMP(k)
#endif

template <bool UNI, bool GI>
__global__ void __launch_bounds__(TRAV_THREADS, 8) k_traverse(TraverseArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int T = (int)a.f.n_trees;

    // XCD-aware block -> (frame, tile): blocks b and b+8 share an XCD (and its L2), so one XCD
    // walks whole frames and the overlapping tile halos of a frame are re-read from one L2.
    // The grid is (8, tiles, frames / 8): the linear workgroup id -- which the hardware deals out to the
    // XCDs round-robin -- is x + 8 * (tile + tiles * z), so no division is needed to decode it.
    int frame = (int)blockIdx.z * 8 + (int)blockIdx.x;
    int tile = (int)blockIdx.y;
    if (a.tile_list) {
        // (k_tile_list) this workgroup takes the k-th flagged tile of the frames = x mod 8; both loads at once
        const uint32_t k = blockIdx.y + gridDim.y * blockIdx.z;
        const uint32_t e = a.tile_list[(size_t)blockIdx.x * a.tile_list_stride + k], cnt = a.tile_list_count[blockIdx.x];
        if (k >= cnt) return;
        frame = (int)(e >> 16) * 8 + (int)blockIdx.x; tile = (int)(e & 0xffffu);
    }
    if (frame >= a.n_frames || KNOB_STOP(a.stop_phase == 9)) return;
    const int ty = div_small(tile, a.tiles_x, 1.0f / (float)a.tiles_x), tx = tile - ty * a.tiles_x;
    const int cx = min(a.px, a.nx - tx * a.px), cy = min(a.py, a.ny - ty * a.py);
    const float r_cx = 1.0f / (float)cx;
    const int npt = cx * cy;
    const int fx0 = tx * a.px * a.step, fy0 = ty * a.py * a.step;   // footprint origin (pixels)
    const int fw = (cx - 1) * a.step + a.sw, fh = (cy - 1) * a.step + a.sh;
    const int ss = a.ss_row;

    uint32_t *sat = lds;
    uint32_t *active = sat + a.ss_max;         // [px * py] window (inside the tile) of every active slot
    uint32_t *agp = active + a.px * a.py;      // [px * py] its position in the frame's window grid
    uint32_t *misc = agp + a.px * a.py;        // [0] n_active, [4] any pixel
    uint32_t *top = misc + 16;                 // (UNI && GI) the tree tops of walk_absorb

    const uint16_t *img = a.frames + (size_t)frame * a.w * a.h;
#ifdef DH_PROFILING_KNOBS
    unsigned long long t_prev = a.dbg_stamps ? clock64() : 0ull;
#endif

    // ---- phase 1.  General path: summed-area table of the footprint, modulo 2^32 (footprints of up
    // to 128 x 128 and 256 x 64 pixels are scanned in registers with DPP wave scans, anything else
    // takes the pass-based build).  Uniform path: copy the tile's region of the frame's box-sum image
    // (k_boxsum): cell (y, x) = sum of the rw x rh rectangle whose top-left pixel is (fx0 + x, fy0 + y).
    // this thread's window (one per thread: npt <= 1024)
    const int wy = div_small(tid, cx, r_cx), wx = tid - wy * cx;
    // k_boxsum (uniform path) / k_pixflags (general path) flagged the tiles whose region holds a non-zero
    // rectangle sum / whose footprint holds a non-zero pixel; any other tile has only background windows
    // (prediction.rs:567-576) and leaves before building anything
    bool nonzero = a.tile_list != nullptr || a.tile_flags[(size_t)frame * (a.tiles_x * a.tiles_y) + tile] != 0;
    if (tid < 8) misc[tid] = 0;
    if (!UNI) __syncthreads();        // (the uniform path's copy ends with a barrier before misc is used)
    if (UNI && nonzero) {
        // The frame's box-sum image is stored with the same column de-interleave as the LDS region
        // (row = m planes of box_plane words; plane c holds the columns = c mod m), so a region row is
        // m runs of q consecutive words and is copied by ONE direct-to-LDS load of 16 bytes per lane
        // (no registers, no LDS-store instructions) whenever the tile's first column sits on a
        // 16-byte boundary of its plane -- the host picks px % 4 == 0 for that.
        const int bh = fh - a.rh + 1, bw = fw - a.rw + 1;
        const int m = 1 << a.swz_log2, q = a.swz_q, X0 = fx0 >> a.swz_log2;      // fx0 is a multiple of m
        const uint32_t *bx = a.box + ((size_t)frame * a.box_rows + fy0) * ((size_t)a.box_plane << a.swz_log2) + X0;
        const int pieces = (m * q) >> 2;                                          // 16-byte pieces per region row
        if ((X0 & 3) == 0 && pieces <= WAVE) {
            const int q4 = q >> 2;
            const int c = (lane * (65536 / q4 + 1)) >> 16, g = lane - c * q4;      // piece -> (plane, group); exact for lane < 64
            const uint32_t *src = bx + (size_t)c * a.box_plane + 4 * g;
            const size_t row_pitch = (size_t)a.box_plane << a.swz_log2;
            // a tile cut off by the frame's edge copies only the groups that hold its own columns (the planes keep
            // their stride q in LDS): the image has just 4 words of slack behind its last column
            const int q4_tile = (((bw + m - 1) >> a.swz_log2) + 3) >> 2;
            if (lane < pieces && g < q4_tile)
                for (int y = tid >> 6; y < bh; y += TRAV_WAVES)
                    __builtin_amdgcn_global_load_lds(src + (size_t)y * row_pitch, sat + y * ss, 16, 0, 0);
        } else {
            const int qd = TRAV_THREADS / bw, rd = TRAV_THREADS - qd * bw;
            int yy = tid / bw, xx = tid - yy * bw;
            const int mm = m - 1;
            while (yy < bh) {
                const int xg = fx0 + xx;
                sat[yy * ss + (xx & mm) * q + (xx >> a.swz_log2)] =
                    a.box[((size_t)frame * a.box_rows + fy0 + yy) * ((size_t)a.box_plane << a.swz_log2) + (size_t)(xg & mm) * a.box_plane + (xg >> a.swz_log2)];
                yy += qd; xx += rd;
                if (xx >= bw) { xx -= bw; ++yy; }
            }
        }
        if (GI) {
            const int tw = T * (1 << a.top_levels) * 3;
            for (int i = tid; i < tw; i += TRAV_THREADS) top[i] = a.top_tab[i];
        }
        __syncthreads();
    }
    if (!UNI && nonzero) {
        const int strip = (fh + TRAV_WAVES - 1) / TRAV_WAVES;
        // (the register-scan build parks 16 strip bottoms of 64 * PPL words in the SAT area before the SAT is written:
        // tiny footprints whose SAT is smaller than that take the pass-based build)
        if (fw <= 2 * WAVE && strip <= 8 && a.ss_max >= TRAV_WAVES * WAVE * 2) nonzero = sat_rows_dpp<2>(sat, &misc[4], img, a.w, fx0, fy0, fw, fh, ss);
        else if (fw <= 4 * WAVE && strip <= 4 && a.ss_max >= TRAV_WAVES * WAVE * 4) nonzero = sat_rows_dpp<4>(sat, &misc[4], img, a.w, fx0, fy0, fw, fh, ss);
        else nonzero = sat_passes(sat, &misc[4], img, a.w, fx0, fy0, fw, fh, ss);
    }
    if (!nonzero) {
        // every pixel under this tile is zero (uniform path: every rectangle sum, and the rectangles
        // cover every window): all of its windows are background (prediction.rs:567-576)
        if (a.dbg_flags)
            for (int p = tid; p < npt; p += TRAV_THREADS) {
                int gp = (ty * a.py + p / cx) * a.nx + tx * a.px + p % cx;
                size_t o = (size_t)frame * a.nx * a.ny + gp;
                a.dbg_flags[o] = 0;
                if (a.dbg_leaf)
                    for (int t = 0; t < T; ++t) a.dbg_leaf[o * T + t] = -1;
            }
        return;
    }
    if (KNOB_STOP(a.stop_phase == 1)) return;
    STAMP(0)

    // ---- phase 2: background gate (prediction.rs:567-571).  Active windows are appended to the tile's
    // segment of the frame's window list (a fixed px * py slots per tile, so no global counter is needed).
    const size_t wbase = (size_t)tile * (a.px * a.py);   // slot of this tile's first window
    uint32_t *wpatch = a.win_patch + (size_t)frame * a.win_cap + wbase;
    if (tid < npt) {                                                   // npt <= 1024: one window per thread
        const int pxi = wx, pyi = wy;
        int ox = pxi * a.step, oy = pyi * a.step;                      // patch origin inside the footprint
        uint32_t sum;
        if (UNI) {
            // the window is covered by rw x rh rectangles at offsets 0, rw, 2rw, ... (the last one
            // clamped to sw - rw); pixel values are non-negative, so the window sum is zero exactly
            // when every one of those rectangle sums is
            // (window origins are multiples of m, so slot(origin + cover offset) = base(origin) + offset:
            // the offsets are uniform and only the base is per lane)
            sum = 0;
            const uint32_t *wp = sat + oy * ss + (ox >> a.swz_log2);
            const int ncx = (a.sw + a.rw - 1) / a.rw, ncy = (a.sh + a.rh - 1) / a.rh;
            for (int iy = 0; iy < ncy; ++iy) {
                const int ro = min(iy * a.rh, a.sh - a.rh) * ss;
                for (int ix = 0; ix < ncx; ix += 4) {               // four reads in flight
                    uint32_t v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int xc = min(min(ix + k, ncx - 1) * a.rw, a.sw - a.rw);
                        v[k] = wp[ro + (xc & ((1 << a.swz_log2) - 1)) * a.swz_q + (xc >> a.swz_log2)];
                    }
                    sum |= v[0] | v[1] | v[2] | v[3];
                }
            }
        } else {
            sum = sat[(oy + a.sh) * ss + ox + a.sw] - sat[oy * ss + ox + a.sw] - sat[(oy + a.sh) * ss + ox] + sat[oy * ss + ox];
        }
        const bool nonbg = sum != 0;   // (sum as f64)/(count as f64) > 0.0  <=>  sum > 0
        const int gp = (ty * a.py + pyi) * a.nx + tx * a.px + pxi;     // position in the frame's window grid
        // slot = running count of active windows: one LDS atomic per wave, ranks from the ballot
        const unsigned long long bal = __ballot(nonbg);
        uint32_t wave_base = 0;
        if (lane == 0 && bal) wave_base = atomicAdd(&misc[0], (uint32_t)__popcll(bal));
        wave_base = __shfl(wave_base, 0);
        if (nonbg) {
            const uint32_t slot = wave_base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            active[slot] = (uint32_t)tid;
            agp[slot] = (uint32_t)gp;
        }
        if (a.dbg_flags) {
            a.dbg_flags[(size_t)frame * a.nx * a.ny + gp] = nonbg ? 1 : 0;
            if (!nonbg && a.dbg_leaf)
                for (int t = 0; t < T; ++t) a.dbg_leaf[((size_t)frame * a.nx * a.ny + gp) * T + t] = -1;
        }
    }
    __syncthreads();
    if (KNOB_STOP(a.stop_phase == 3)) return;
    STAMP(1)
    const int n_active = (int)misc[0];
    if (tid == 0) a.win_count[(size_t)frame * (a.tiles_x * a.tiles_y) + tile] = (uint32_t)n_active;   // zero for skipped tiles (host memset)
    if (n_active == 0) return;     // nothing to walk (debug taps were written above)
    // the window list's positions: stored after the barrier so that nothing waits for the stores
    if (tid < n_active) wpatch[tid] = agp[tid];

    // ---- phase 3: root->leaf walks.  Work item k = (tree k / n_active, active slot k % n_active),
    // lane = k mod 1024: the lanes of a wave walk the SAME tree for NEIGHBOURING windows (4 px apart),
    // which see almost the same pixels, so they mostly follow the same path: node fetches collapse
    // to a few addresses per wave.  (On the bench forest a third of the walks reach the maximum depth and the rest end
    // anywhere above it, so a wave's chain runs the full depth while its lanes need half of it.)
    // (Measured and rejected on MI355X, DESIGN.md section 4: a ballot-compacted refill of finished
    // lanes lengthens this phase by 25-50 % because it breaks exactly that coherence; a persistent launch
    // with per-XCD tile queues gains nothing.  An LDS copy of the tree tops lost in round 1, when it displaced
    // tile area under a slower loop; as an 8-byte-per-slot implicit heap it is what walk_absorb uses now.)
    // Trees are validated acyclic on the host, so every walk ends.
    const int total = n_active * T;
    int32_t *wleaf = a.win_leaf + (size_t)frame * a.win_cap * T + wbase;                     // [tree][win_cap] per frame
    int32_t *dleaf = a.dbg_leaf ? a.dbg_leaf + (size_t)frame * a.nx * a.ny * T : nullptr;
    if (UNI) {
        // W walks per lane, advanced in lock step: their node fetches and box-sum reads are
        // independent, so each lane keeps W dependent-load chains in flight.  W = walks per lane
        // of this tile (at most 4), so one pass covers the tile whenever it has <= 4096 walks.
        // (per WAVE: the last pass over the items is usually filled in part, and a wave without items in it walks one chain
        // fewer -- every lock-step level of a chain is a node gather of the whole wave, whatever its lanes hold)
        const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~(WAVE - 1));
        const int all_passes = (total - wave_first + TRAV_THREADS - 1) / TRAV_THREADS;
        const int per_lane = all_passes > 4 ? (total + TRAV_THREADS - 1) / TRAV_THREADS : all_passes;
        if (GI) {                   // (uniform path: the second template flag selects the absorbing-leaf walk table)
            if (per_lane <= 1) walk_absorb<1>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
            else if (per_lane == 2) walk_absorb<2>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
            else if (per_lane == 3) walk_absorb<3>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
            else walk_absorb<4>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
        } else
        if (per_lane <= 1) walk_uniform<1>(a, sat, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
        else if (per_lane == 2) walk_uniform<2>(a, sat, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
        else if (per_lane == 3) walk_uniform<3>(a, sat, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
        else walk_uniform<4>(a, sat, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
    } else if (GI) {
        const int per_lane = (total + TRAV_THREADS - 1) / TRAV_THREADS;
        if (per_lane <= 1) walk_general_int<1>(a, sat, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
        else walk_general_int<2>(a, sat, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
    } else {
        for (int k = tid; k < total; k += TRAV_THREADS) {
            const int t = div_small(k, n_active, 1.0f / (float)n_active), slot = k - t * n_active;
            const int p = (int)active[slot];
            const int pyi = div_small(p, cx, r_cx), pxi = p - pyi * cx;
            const uint32_t *sp = sat + pyi * a.step * ss + pxi * a.step;
            int cur = a.f.roots[t];
            while (cur >= 0) {
                // general rectangles: 8 SAT corners, IEEE f64 means (types.rs:317-339)
                const uint4 *np = (const uint4 *)(a.f.nodes + cur);
                const uint4 n0 = np[0], n1 = np[1];
                const int ax0 = n0.x & 0xffff, ay0 = n0.x >> 16, ax1 = n0.y & 0xffff, ay1 = n0.y >> 16;
                const int bx0 = n0.z & 0xffff, by0 = n0.z >> 16, bx1 = n0.w & 0xffff, by1 = n0.w >> 16;
                const double thr = __hiloint2double((int)n1.y, (int)n1.x);
                const uint32_t s1 = sp[ay1 * ss + ax1] - sp[ay0 * ss + ax1] - sp[ay1 * ss + ax0] + sp[ay0 * ss + ax0];
                const uint32_t s2 = sp[by1 * ss + bx1] - sp[by0 * ss + bx1] - sp[by1 * ss + bx0] + sp[by0 * ss + bx0];
                const uint32_t c1 = (uint32_t)((ax1 - ax0) * (ay1 - ay0)), c2 = (uint32_t)((bx1 - bx0) * (by1 - by0));
                const double a1 = c1 ? __ddiv_rn((double)s1, (double)c1) : 0.0;   // types.rs:335-338
                const double a2 = c2 ? __ddiv_rn((double)s2, (double)c2) : 0.0;
                cur = (__dsub_rn(a1, a2) > thr) ? (int)n1.w : (int)n1.z;
            }
            wleaf[(size_t)t * a.win_cap + slot] = ~cur;
            if (dleaf) dleaf[(size_t)agp[slot] * T + t] = ~cur;
        }
    }

    STAMP(2)
}

// ================================================================== k_emit
// One thread per active window of the list k_traverse wrote: mean leaf probability in tree order
// (prediction.rs:582-584), the > 0.7 gate, and one self-contained hit record per voting leaf
// (consumed by k_vote and k_cluster).  No LDS, no barriers: the three dependent global round trips
// (leaf probabilities, the frame's hit counter, the leaf templates) that used to end every
// k_traverse workgroup are hidden here by plain occupancy.
#define EMIT_THREADS 256
// EB = trees handled per batch of gathers: the smallest instance that holds all T trees keeps the registers (and with them the
// occupancy of this latency-bound kernel) in proportion to the forest: 62 VGPRs at EB = 8, 80 at 10, 122 at 16.
template <int EB>
__global__ void __launch_bounds__(EMIT_THREADS) k_emit(EmitArgs a) {
    const int frame = blockIdx.y, lane = threadIdx.x & (WAVE - 1);
    const int pp = a.px * a.py;
    // Thread i of the frame takes the i-th active window in tile order: every wave scans the tiles'
    // counts itself (64 tiles per step, one load per lane) and finds its tile with a binary search
    // over the running totals, so only the blocks past the frame's last active window are idle.
    const uint32_t *counts = a.win_count + (size_t)frame * a.tiles;
    const uint32_t idx = (uint32_t)blockIdx.x * EMIT_THREADS + threadIdx.x;
    uint32_t before = 0;                 // active windows in the tiles of earlier steps
    int tile = -1, slot = 0;
    for (int t0 = 0; t0 < a.tiles; t0 += WAVE) {
        const uint32_t c = t0 + lane < a.tiles ? counts[t0 + lane] : 0u;
        const uint32_t inc = wave_incl_scan(c);
        const uint32_t step_total = (uint32_t)__shfl((int)inc, WAVE - 1);
        // (all lanes run the search: a shuffle must not read a lane that sits out a divergent branch)
        const uint32_t r = idx - before;                                  // rank inside this step, if the window is in it
        int s = 0;                                                        // = number of lanes whose running total is <= r
#pragma unroll
        for (int b = WAVE / 2; b; b >>= 1)
            if ((uint32_t)__shfl((int)inc, s + b - 1) <= r) s += b;
        s = min(s, WAVE - 1);
        const uint32_t first = (uint32_t)__shfl((int)inc, s) - (uint32_t)__shfl((int)c, s);
        if (tile < 0 && idx >= before && r < step_total) { tile = t0 + s; slot = (int)(r - first); }
        before += step_total;
        if (__ballot(tile < 0) == 0ull) break;
    }
    const bool live = tile >= 0;
    if (__ballot(live) == 0ull || KNOB_STOP(a.stop == 1)) return;
    if (!live) { tile = 0; slot = 0; }
    const int T = (int)a.f.n_trees;
    const size_t w = (size_t)tile * pp + slot;                           // slot in the frame's window list
    uint32_t cnt = 0, gp = 0;
    bool gated = false;
    const int32_t *wl = a.win_leaf + (size_t)frame * a.win_cap * T + w;
    unsigned long long voting = 0;          // bit t: the leaf reached in tree t casts votes (T <= 64; else recomputed below)
    uint32_t rotv = 0, l[EB];       // bit k: leaf l[k] casts rotation votes and the window passed the gate
#pragma unroll
    for (int k = 0; k < EB; ++k) l[k] = 0;
    uint16_t zc = 0;                        // depth at the window centre
    if (live) {
        gp = a.win_patch[(size_t)frame * a.win_cap + w];
        // Batches of EB trees: the leaf ids, then their probabilities and flags, are requested
        // together, so a window costs two dependent round trips per batch (one batch for T <= 16).
        double prob = 0.0;
        for (int t0 = 0; t0 < T; t0 += EB) {
            uint32_t lf[EB];
            double pr[EB];
            uint4 g[EB];
#pragma unroll
            for (int k = 0; k < EB; ++k) l[k] = (uint32_t)wl[(size_t)min(t0 + k, T - 1) * a.win_cap];
            if (t0 == 0) {
                // window centre (for prediction.rs:551-554), requested now: its latency hides behind the batch
                const int gyi = (int)(gp / (uint32_t)a.nx), gxi = (int)gp - gyi * a.nx;
                zc = a.frames[(size_t)frame * a.w * a.h + (size_t)(gyi * a.step + a.lh) * a.w + gxi * a.step + a.lw];
            }
#pragma unroll
            for (int k = 0; k < EB; ++k) g[k] = ((const uint4 *)(a.f.tpl + l[k]))[3];             // n_rot, flags, prob
#pragma unroll
            for (int k = 0; k < EB; ++k) { lf[k] = g[k].y; pr[k] = __hiloint2double((int)g[k].w, (int)g[k].z); }
#pragma unroll
            for (int k = 0; k < EB; ++k)
                if (t0 + k < T) {
                    prob = __dadd_rn(prob, pr[k]);                                   // tree order, f64 (prediction.rs:582-584)
                    if ((lf[k] & LF_PROB) && (lf[k] & (LF_ROT | LF_OFF))) {
                        cnt++; voting |= 1ull << ((t0 + k) & 63);
                        if (lf[k] & LF_ROT) rotv |= 1u << k;                          // (only read when T <= EB)
                    }
                }
        }
        prob = __ddiv_rn(prob, (double)T);
        gated = prob > DH_PROB_GATE;
        if (!gated) { cnt = 0; voting = 0; rotv = 0; }
        if (gated && a.dbg_flags) a.dbg_flags[(size_t)frame * a.npatch + gp] = 3;
    }
    // Leaf histogram (rotation votes per leaf, read by k_vote and k_cluster): neighbouring windows -- adjacent
    // lanes -- mostly reach the same leaf of a tree, so runs of equal leaves along the wave are counted with
    // one ballot and added by the run's first lane: a few times fewer global atomics than one per hit record.
    const bool hist_here = a.leaf_hits && T <= EB;
    if (hist_here) {
        uint32_t *lhist = a.leaf_hits + (size_t)frame * a.f.n_leaves;
#pragma unroll
        for (int k = 0; k < EB; ++k) {
            if (k >= T) continue;                                                    // (uniform)
            const bool v = (rotv >> k) & 1u;
            const uint32_t key = v ? l[k] : 0xFFFFFFFFu;
            const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
            const bool cont = v && lane > 0 && prev == key;                          // continues the previous lane's run
            const unsigned long long c = __ballot(cont);
            if (v && !cont) {
                const unsigned long long rest = lane == WAVE - 1 ? 0ull : (c >> (lane + 1));
                atomicAdd(&lhist[key], 1u + (uint32_t)__builtin_ctzll(~rest));      // run length = 1 + following continuations
            }
        }
    }
    if (KNOB_STOP(a.stop == 2)) return;
    // slots in the frame's hit arrays: one atomic per wave, exclusive prefix of the lanes' counts
    const uint32_t incl = wave_incl_scan(cnt);
    const uint32_t wave_total = __shfl(incl, WAVE - 1);
    if (wave_total == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.hit_count[frame], wave_total);
    base = __shfl(base, 0);
    const uint32_t excl = incl - cnt;
    // window centre -> 3-D (prediction.rs:551-554), by the lanes whose window votes
    float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
    if (cnt) {
        const int gyi = (int)(gp / (uint32_t)a.nx), gxi = (int)gp - gyi * a.nx;
        float q[3];
        to3d(a.kinv, (float)(gxi * a.step + a.lw), (float)(gyi * a.step + a.lh), (float)zc, q);
        q0 = q[0]; q1 = q[1]; q2 = q[2];
    }
    HitRec *dst = a.hits + (size_t)frame * a.hits_cap;
    HitBox *dbox = a.hit_box + (size_t)frame * a.hits_cap;
    HitRot *drot = a.hit_rot + (size_t)frame * a.hits_cap;
    const int32_t *wlf = a.win_leaf + (size_t)frame * a.win_cap * T;
    const int wi = (int)w;
    // One LANE per hit record: hit h of the wave belongs to the voting window (lane) s with
    // excl_s <= h < excl_s + cnt_s and is its (h - excl_s)-th voting tree.  Every lane finds its (s, tree)
    // with two six-step binary searches (over the prefix counts, then over the voting mask), and all
    // records of the wave are built at once: one leaf-id load, one template load, one store per lane
    // instead of a per-window loop over the trees.
    for (uint32_t chunk = 0; chunk < wave_total; chunk += WAVE) {
        int src = 0, tree = 0;
        if (T <= 64) {
            // src = number of lanes whose inclusive count is <= h (the counts are non-decreasing)
            const uint32_t h = chunk + (uint32_t)lane;
#pragma unroll
            for (int b = WAVE / 2; b; b >>= 1)
                if ((uint32_t)__shfl((int)incl, src + b - 1) <= h) src += b;
            src = min(src, WAVE - 1);                                     // lanes past the last hit: any valid source
            uint32_t n = h - (uint32_t)__shfl((int)excl, src);            // rank of the hit among its window's voting trees
            const uint32_t vlo = (uint32_t)__shfl((int)(uint32_t)voting, src), vhi = (uint32_t)__shfl((int)(uint32_t)(voting >> 32), src);
            // tree = position of the n-th set bit of the window's voting mask
            uint32_t word = vlo;
            const uint32_t clo = (uint32_t)__popc(vlo);
            if (n >= clo) { n -= clo; word = vhi; tree = 32; }
#pragma unroll
            for (int b = 16; b; b >>= 1) {
                const uint32_t c = (uint32_t)__popc(word & ((1u << b) - 1u));
                if (n >= c) { n -= c; word >>= b; tree += b; }
            }
        } else {
            // more than 64 trees: every lane finds its hit by walking the voting lanes' leaves itself
            const uint32_t h = chunk + (uint32_t)lane;
            for (int sl = 0; sl < WAVE; ++sl) {
                const uint32_t ex = (uint32_t)__shfl((int)excl, sl), cn = (uint32_t)__shfl((int)cnt, sl);
                const int sw = __shfl(wi, sl);
                if (h < ex || h >= ex + cn) continue;
                uint32_t r = ex;
                for (int t = 0; t < T; ++t) {
                    const uint32_t lf = a.f.leaf_flags[(uint32_t)wlf[(size_t)t * a.win_cap + sw]];
                    if ((lf & LF_PROB) && (lf & (LF_ROT | LF_OFF))) { if (r == h) { src = sl; tree = t; } ++r; }
                }
            }
        }
        const uint32_t h = chunk + (uint32_t)lane;
        const int sw = __shfl(wi, src);
        const float p0 = __shfl(q0, src), p1 = __shfl(q1, src), p2 = __shfl(q2, src);
        const uint32_t o = base + h;
        if (h < wave_total && o < a.hits_cap) {
            const uint32_t lid = (uint32_t)wlf[(size_t)tree * a.win_cap + sw];
            const uint4 *tp = (const uint4 *)(a.f.tpl + lid);
            const uint4 t0 = tp[0], t1 = tp[1], t2v = tp[2], t3 = tp[3];
            const float mn0 = __uint_as_float(t0.x), mn1 = __uint_as_float(t0.y), mn2 = __uint_as_float(t0.z),
                        mx0 = __uint_as_float(t0.w), mx1 = __uint_as_float(t1.x), mx2 = __uint_as_float(t1.y);
            *(float4 *)(dst + o) = make_float4(p0, p1, p2, __uint_as_float(t2v.x));              // p3, ob
            ((int4 *)(dbox + o))[0] = make_int4(f32_as_i32(__fsub_rn(p0, mx0)), f32_as_i32(__fsub_rn(p1, mx1)),
                                                 f32_as_i32(__fsub_rn(p2, mx2)), f32_as_i32(__fsub_rn(p0, mn0)));
            ((int4 *)(dbox + o))[1] = make_int4(f32_as_i32(__fsub_rn(p1, mn1)), f32_as_i32(__fsub_rn(p2, mn2)), (int)t1.z, (int)t1.w);
            // the rotation record is only read when there is no leaf histogram (k_vote, k_cluster) or by the vote-dump tap
            if (!a.leaf_hits || a.dbg_flags) *(uint4 *)(drot + o) = make_uint4(t2v.y, t2v.z, t2v.w, t3.x);    // rlo, rhi, rb, n_rot
            if (a.leaf_hits && !hist_here && (t1.w & LF_ROT)) atomicAdd(&a.leaf_hits[(size_t)frame * a.f.n_leaves + lid], 1u);
        }
    }
}

hipError_t dh_launch_emit(const EmitArgs &a, hipStream_t s) {
    if (a.n_frames == 0 || a.tiles == 0 || a.npatch == 0) return hipSuccess;
    if (a.n_frames > 65535) return hipErrorInvalidConfiguration;
    const dim3 grid((a.npatch + EMIT_THREADS - 1) / EMIT_THREADS, a.n_frames), block(EMIT_THREADS);
    const uint32_t T = a.f.n_trees;
    if (T <= 4) hipLaunchKernelGGL(k_emit<4>, grid, block, 0, s, a);
    else if (T <= 8) hipLaunchKernelGGL(k_emit<8>, grid, block, 0, s, a);
    else if (T <= 10) hipLaunchKernelGGL(k_emit<10>, grid, block, 0, s, a);
    else if (T <= 12) hipLaunchKernelGGL(k_emit<12>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k_emit<16>, grid, block, 0, s, a);
    return hipGetLastError();
}

// Raise the dynamic-LDS limit of the walk kernel (once per process; not allowed during stream capture,
// so dh_predictor_create calls it).
hipError_t dh_kernels_init() {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_traverse<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_traverse<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_traverse<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_traverse<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    return hipSuccess;
}

hipError_t dh_launch_traverse(const TraverseArgs &a, size_t lds_bytes, hipStream_t s) {
    const int tiles = a.tiles_x * a.tiles_y, fb = (a.n_frames + 7) / 8;
    if (tiles == 0 || fb == 0) return hipSuccess;
    if (tiles > 65535 || fb > 65535) return hipErrorInvalidConfiguration;
    const dim3 grid(8, tiles, fb);
    if (a.uniform && a.nodes_a) hipLaunchKernelGGL((k_traverse<true, true>), grid, dim3(TRAV_THREADS), lds_bytes, s, a);
    else if (a.uniform) hipLaunchKernelGGL((k_traverse<true, false>), grid, dim3(TRAV_THREADS), lds_bytes, s, a);
    else if (a.nodes_g) hipLaunchKernelGGL((k_traverse<false, true>), grid, dim3(TRAV_THREADS), lds_bytes, s, a);
    else hipLaunchKernelGGL((k_traverse<false, false>), grid, dim3(TRAV_THREADS), lds_bytes, s, a);
    return hipGetLastError();
}

// ================================================================== k_vote
// Coarse guess grids (prediction.rs:529-533, :630-636, :661-676).  Each workgroup owns a slice of
// one frame's hit records, accumulates in LDS and flushes its non-zero cells with integer atomics
// (exact, order-free, wrapping like the reference's release-mode u32 `+=`).
#define VOTE_THREADS 512
#ifndef VOTE_SLICES
#define VOTE_SLICES 8
#endif
#define VOTE_TAB 2048
#ifndef VOTE_SUB
#define VOTE_SUB 4u              // lanes that share one hit record (measured on MI355X: 4 x 12 beats 8 x 6 by 9 %, 2 x 12 by 4 %)
#endif
#ifndef VOTE_ILP
#define VOTE_ILP 12              // offset votes a lane keeps in flight (4 lanes x 12 = a 48-vote leaf in one round)
#endif

// Position votes of one hit record into the workgroup's 20 x 20 grid (prediction.rs:647-676): lane `sub` of the VOTE_SUB
// lanes sharing the record takes the leaf's votes sub, sub + VOTE_SUB, ...  PINNED: pinhole form of the projection (k_vote).
template <bool TAB, bool PINNED>
__device__ __forceinline__ void vote_positions(const VoteArgs &a, uint32_t *pos, const uint8_t *gxt, const uint8_t *gyt, const float4 rec,
                                               uint32_t v, uint32_t fc, uint32_t sub, float wm1, float hm1) {
    const uint32_t ob = __float_as_uint(rec.w), oe = ob + (fc >> 8);
    uint32_t last = 0xFFFFFFFFu, acc = 0;      // neighbouring votes mostly share a cell: one atomic per run
    for (uint32_t o0 = ob + sub; o0 < oe; o0 += VOTE_SUB * VOTE_ILP) {   // the lane's next VOTE_ILP votes: loads first
        float ox[VOTE_ILP], oy[VOTE_ILP], oz[VOTE_ILP];
#pragma unroll
        for (int j = 0; j < VOTE_ILP; ++j) {
            const uint32_t o = min(o0 + VOTE_SUB * j, oe - 1);
            const float4 of = a.f.off4[o];                          // 4 lanes x 16 B = one 64-byte line per record group
            ox[j] = of.x; oy[j] = of.y; oz[j] = of.z;
        }
#pragma unroll
        for (int j = 0; j < VOTE_ILP; ++j) {
            if (o0 + VOTE_SUB * j >= oe) break;
            float nx = __fsub_rn(rec.x, ox[j]), ny = __fsub_rn(rec.y, oy[j]), nz = __fsub_rn(rec.z, oz[j]); // :647
            if (nz < 0.0f) continue;                                              // :650
            float r[3];
            if (PINNED) {
                r[0] = __fadd_rn(__fmul_rn(nx, a.k[0]), __fmul_rn(nz, a.k[2]));
                r[1] = __fadd_rn(__fmul_rn(ny, a.k[4]), __fmul_rn(nz, a.k[5]));
                r[2] = __fadd_rn(nz, 0.0f);                                       // (x * 0 + y * 0) + z * 1: -0 becomes +0
            } else {
                matvec3(a.k, nx, ny, nz, r);                                      // types.rs:425
            }
            // Only the CELL of the 20 x 20 grid is needed here.  When w and h are multiples of 20 (cells are whole pixels wide)
            // the cell of the reference's clamped, truncated quotient x2 (:662-672) is floor(clamp(x2 * 20 / w)), and an
            // approximate quotient decides it whenever it is not next to a cell border: with v_rcp_f32 (1 ulp) and three more
            // roundings u~ = (r0 * rcp(r2)) * (20 / w) is within 2.5 * 2^-23 * |u| < 7e-6 of the real u for |u| <= 21, the
            // reference's own rounding of r0 / r2 moves it by < 2e-6 more, and outside [0, 20) both sides clamp into cell 0 / 19
            // wherever they are.  Quotients within 1e-4 of an integer, and everything not finite (r2 = 0, NaN), take the two
            // IEEE divisions -- 0.04 % of the votes.  Saves 22 of the 52 VALU instructions of a vote.
            uint32_t idx;
            const float rc = __builtin_amdgcn_rcpf(r[2]);
            const float ux = __fmul_rn(__fmul_rn(r[0], rc), a.sx), uy = __fmul_rn(__fmul_rn(r[1], rc), a.sy);
            const bool near_border = !(fabsf(__fsub_rn(ux, rintf(ux))) > 1.0e-4f) || !(fabsf(__fsub_rn(uy, rintf(uy))) > 1.0e-4f);
            if (a.cell_fast && !near_border) {
                const float cxf = fminf(fmaxf(ux, 0.0f), 19.5f), cyf = fminf(fmaxf(uy, 0.0f), 19.5f);
                idx = (uint32_t)cyf * DH_GRID + (uint32_t)cxf;
            } else {
                float qx = __fdiv_rn(r[0], r[2]), qy = __fdiv_rn(r[1], r[2]);
                float x2 = qx > 0.0f ? qx : 0.0f; x2 = x2 < wm1 ? x2 : wm1;           // :662
                float y2 = qy > 0.0f ? qy : 0.0f; y2 = y2 < hm1 ? y2 : hm1;           // :663
                // x2 in [0, w-1] and never NaN after the clamps: `as usize` is a plain truncation
                const uint32_t xi = (uint32_t)x2, yi = (uint32_t)y2;
                const uint32_t gx = TAB ? gxt[xi] : xi * DH_GRID / (uint32_t)a.w;      // :671-672
                const uint32_t gy = TAB ? gyt[yi] : yi * DH_GRID / (uint32_t)a.h;
                idx = gy * DH_GRID + gx;
            }
            if (idx != last) {
                if (acc) atomicAdd(&pos[last], acc);                              // :675
                last = idx; acc = 0;
            }
            acc += v;
        }
    }
    if (acc) atomicAdd(&pos[last], acc);
}

// PIN: the intrinsic matrix has the pinhole form [[fx, 0, cx], [0, fy, cy], [0, 0, 1]] (types.rs:418-420 and every BIWI
// calibration): of the nine products of space_to_img_coord's matrix-vector product (types.rs:425, meancov_estimation.rs:201-216)
// five are x * 0 or z * 1.  For finite operands they are exact no-ops -- a + (+-0) = a, and the sign of a zero sum only
// matters for r2, restored by adding +0 -- so four products and three sums give bit-identical r.  Hits whose window centre or
// leaf offsets are not finite and small (LF_FIN) take the general expression (there 0 * inf = NaN must propagate).
template <bool TAB, bool PIN>
__global__ void __launch_bounds__(VOTE_THREADS) k_vote(VoteArgs a) {
    __shared__ uint32_t pos[DH_POSGRID];
    __shared__ uint32_t rot[DH_GRID3];
    __shared__ uint8_t gxt[VOTE_TAB], gyt[VOTE_TAB];   // pixel -> guess-grid column / row: x * 20 / w (:671-674) without a division per vote
    const int frame = blockIdx.y, tid = threadIdx.x;
    uint32_t n = a.hit_count[frame];
    if (n > a.hits_cap) n = a.hits_cap;
    const uint32_t slices = gridDim.x;
    const uint32_t per = (n + slices - 1) / slices;
    const uint32_t h0 = min(n, blockIdx.x * per), h1 = min(n, h0 + per);
    if (n == 0 || (h0 >= h1 && !a.leaf_hits)) return;       // with the leaf histogram every slice also owns a share of the leaves
    for (int i = tid; i < DH_POSGRID; i += VOTE_THREADS) pos[i] = 0;
    for (int i = tid; i < DH_GRID3; i += VOTE_THREADS) rot[i] = 0;
    if (TAB) {
        for (int i = tid; i < a.w; i += VOTE_THREADS) gxt[i] = (uint8_t)((uint32_t)i * DH_GRID / (uint32_t)a.w);
        for (int i = tid; i < a.h; i += VOTE_THREADS) gyt[i] = (uint8_t)((uint32_t)i * DH_GRID / (uint32_t)a.h);
    }
    __syncthreads();
    if (KNOB_STOP(a.stop == 1)) return;
    const HitRec *hits = a.hits + (size_t)frame * a.hits_cap;
    const HitBox *box = a.hit_box + (size_t)frame * a.hits_cap;
    const HitRot *hr = a.hit_rot + (size_t)frame * a.hits_cap;
    const float wm1 = (float)(a.w - 1), hm1 = (float)(a.h - 1);
    // VOTE_SUB lanes share one hit record: lane `sub` takes the leaf's votes sub, sub + VOTE_SUB, ... so the
    // chain of dependent offset loads per lane is n_votes / VOTE_SUB long and all lanes of the workgroup stay busy
    const uint32_t sub = tid & (VOTE_SUB - 1u);
    // the records of a lane's NEXT hit are requested before the current one is worked on: a hit then costs one dependent
    // round trip (its offset votes) instead of two
    const uint32_t i_first = h0 + tid / VOTE_SUB;
    float4 rec_n = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 b1_n = make_int4(0, 0, 0, 0);
    if (i_first < h1) { rec_n = *(const float4 *)(hits + i_first); b1_n = ((const int4 *)(box + i_first))[1]; }
    for (uint32_t i = i_first; i < h1; i += VOTE_THREADS / VOTE_SUB) {
        const float4 rec = rec_n;
        const int4 b1 = b1_n;
        const uint32_t i_next = i + VOTE_THREADS / VOTE_SUB;
        if (i_next < h1) { rec_n = *(const float4 *)(hits + i_next); b1_n = ((const int4 *)(box + i_next))[1]; }
        const uint4 rr = a.leaf_hits ? make_uint4(0u, 0u, 0u, 0u) : *(const uint4 *)(hr + i);   // rotation cells: only without the leaf histogram
        const uint32_t v = (uint32_t)b1.z, fc = (uint32_t)b1.w;
        if ((fc & LF_ROT) && !a.leaf_hits)
            for (uint32_t r = rr.z + sub; r < rr.z + (rr.w >> 16); r += VOTE_SUB) atomicAdd(&rot[a.f.rot_rough[r]], v * a.f.rough_mult[r]);   // :636
        // the pinhole form of the projection is taken by whole waves (a wave with one hit whose operands are not finite and
        // small takes the general expression for all of its hits: a uniform branch, not a per-lane select of both results)
        const bool pin_lane = PIN && (fc & LF_FIN) && fabsf(rec.x) < 1.0e30f && fabsf(rec.y) < 1.0e30f && fabsf(rec.z) < 1.0e30f;
        const bool pin_wave = PIN && __ballot((fc & LF_OFF) && !pin_lane) == 0ull;
        if (fc & LF_OFF) {
            if (pin_wave) vote_positions<TAB, true>(a, pos, gxt, gyt, rec, v, fc, sub, wm1, hm1);
            else vote_positions<TAB, false>(a, pos, gxt, gyt, rec, v, fc, sub, wm1, hm1);
        }
    }
    if (KNOB_STOP(a.stop == 2)) return;
    if (a.leaf_hits) {
        // Rotation votes depend only on the leaf (prediction.rs:601-636): with the per-frame leaf histogram the
        // 20^3 guess grid is the sum over the leaves that voted of count x v x (their distinct cells); u32
        // wrap-around makes that the same residue as count separate adds.  The slices share the leaves.
        const uint32_t *lh = a.leaf_hits + (size_t)frame * a.f.n_leaves;
        for (uint32_t l = blockIdx.x * VOTE_THREADS + tid; l < a.f.n_leaves; l += slices * VOTE_THREADS) {
            const uint32_t c = lh[l];
            if (!c) continue;
            const uint4 *tp = (const uint4 *)(a.f.tpl + l);
            const uint4 t1 = tp[1], t2 = tp[2], t3 = tp[3];
            if (!(t1.w & LF_ROT)) continue;
            const uint32_t cv = c * t1.z;                      // count x valtoadd
            for (uint32_t r = t2.w; r < t2.w + (t3.x >> 16); ++r) atomicAdd(&rot[a.f.rot_rough[r]], cv * a.f.rough_mult[r]);   // :636
        }
    }
    __syncthreads();
    if (KNOB_STOP(a.stop == 3)) return;
    uint32_t *gp = a.pos_grid + (size_t)frame * DH_POSGRID, *gr = a.rot_grid + (size_t)frame * DH_GRID3;
    for (int i = tid; i < DH_POSGRID; i += VOTE_THREADS) if (pos[i]) atomicAdd(&gp[i], pos[i]);
    for (int i = tid; i < DH_GRID3; i += VOTE_THREADS) if (rot[i]) atomicAdd(&gr[i], rot[i]);
}

hipError_t dh_launch_vote(const VoteArgs &a, hipStream_t s) {
    if (a.n_frames == 0) return hipSuccess;
    const bool pin = a.k[1] == 0.0f && a.k[3] == 0.0f && a.k[6] == 0.0f && a.k[7] == 0.0f && a.k[8] == 1.0f;
    // slices per frame: 8 for batches that fill the chip by their frames, more for small batches (a slice flushes at most
    // 8 400 cells with atomics, so 64 slices of one frame still cost less than a mostly idle chip)
    const uint32_t slices = a.n_frames >= 128 ? VOTE_SLICES : std::min(64u, std::max((uint32_t)VOTE_SLICES, 1024u / (uint32_t)a.n_frames));
    const dim3 grid(slices, a.n_frames), block(VOTE_THREADS);
    VoteArgs b = a;
    b.cell_fast = a.cell_fast && a.w % DH_GRID == 0 && a.h % DH_GRID == 0 && a.w > 0 && a.h > 0;
    b.sx = (float)DH_GRID / (float)a.w; b.sy = (float)DH_GRID / (float)a.h;
    if (a.w <= VOTE_TAB && a.h <= VOTE_TAB) {
        if (pin) hipLaunchKernelGGL((k_vote<true, true>), grid, block, 0, s, b);
        else hipLaunchKernelGGL((k_vote<true, false>), grid, block, 0, s, b);
    } else {
        if (pin) hipLaunchKernelGGL((k_vote<false, true>), grid, block, 0, s, b);
        else hipLaunchKernelGGL((k_vote<false, false>), grid, block, 0, s, b);
    }
    return hipGetLastError();
}

// ================================================================== k_cluster
// One 1024-thread workgroup per (frame, accumulator): blockIdx.x = 0 head position (`mid`),
// 1 rotation (`rot`).
//
// The reference keeps both accumulators as unbounded HashMap<(i32,i32,i32),u32>
// (meanshift.rs:14-68) and reads a 20^3 window per iteration.  Here a 26^3-cell REGION of the
// accumulator around the current position is materialised in LDS straight from the hit records
// (integer atomics: exact, order-free); hits whose vote bounding box misses the region are dropped
// with one test.  The mean shift then iterates inside the region and the gather is repeated only
// when the 20^3 window would leave it (it moves by a few cells per step after the first).
// The weighted sums run over the non-zero window cells in the reference's x -> y -> z order
// (meanshift.rs:344-381) as a strictly sequential f32 chain on 4 lanes (num.x, num.y, num.z, den);
// everything off that chain (cell compaction, kernel weight, products) is done by all threads.
#define CL_THREADS 1024
#define CL_WAVES (CL_THREADS / WAVE)
#define CL_CHUNKS 8             // 8 * 1024 = 8192 >= 8000 window cells
#define CL_PROD_CAP 512         // products staged per pass (x4 floats = 8 KB)
#define CL_LIST (CL_PROD_CAP * 4) // survivors of the region gathers' bounding-box tests listed in `prod` (2048)
#define RG 26                   // region edge; the window may sit at offsets 0..RG-20 inside it
#define RG3 (RG * RG * RG)
static_assert(RG3 == DH_REGION_CELLS, "dh_internal.h: DH_REGION_CELLS");

// exists c in [lo,hi] and d in [0,len) with c == start + d (i32 wrapping, like the reference's
// release-mode `pos + offset`)?
__device__ __forceinline__ bool range_hits_span(int32_t lo, int32_t hi, int32_t start, uint32_t len) {
    uint32_t u = (uint32_t)start - (uint32_t)lo;
    return u <= (uint32_t)hi - (uint32_t)lo || u >= (uint32_t)(1u - len);
}

// Position votes of hit record i that fall into the region: lane `sub` of `nsub` takes the leaf's offset votes
// sub, sub + nsub, ... (prediction.rs:647-667).
__device__ __forceinline__ void cluster_add_votes(const ClusterArgs &a, uint32_t *region, const HitRec *hits, const HitBox *box,
                                                  uint32_t i, uint32_t sub, uint32_t nsub, const int32_t org[3]) {
    const float4 rec = *(const float4 *)(hits + i);
    const uint32_t v = box[i].v, fc = box[i].fc;
    const uint32_t ob = __float_as_uint(rec.w), oe = ob + (fc >> 8);
#pragma unroll 1
    for (uint32_t o = ob + sub; o < oe; o += nsub) {
        const float4 of = a.f.off4[o];
        float nx = __fsub_rn(rec.x, of.x), ny = __fsub_rn(rec.y, of.y), nz = __fsub_rn(rec.z, of.z); // prediction.rs:647
        if (nz < 0.0f) continue;                                                                      // :650
        uint32_t dx = (uint32_t)f32_as_i32(nx) - (uint32_t)org[0];                                     // :667
        uint32_t dy = (uint32_t)f32_as_i32(ny) - (uint32_t)org[1];
        uint32_t dz = (uint32_t)f32_as_i32(__fdiv_rn(nz, (float)DH_ZSCALEFACTOR)) - (uint32_t)org[2];
        if (dx < RG && dy < RG && dz < RG) atomicAdd(&region[(dx * RG + dy) * RG + dz], v);
    }
}

// Shared-memory carve-up of k_cluster / k_region.
struct ClShared {
    uint32_t *region;              // [RG3]
    float *prod;                   // [CL_PROD_CAP * 4], doubles as the survivor list of the gathers
    unsigned long long *red64;     // [CL_WAVES]
    uint32_t *red32;               // [CL_WAVES]
    int32_t *s_pos;                // [3]
    uint32_t *s_total;
};

// Initial guess of one accumulator into sh.s_pos (the caller synchronises before reading it): first strictly-greatest
// cell, i.e. greatest value then smallest index (prediction.rs:694-702 for the 20x20 grid; :733-742 with the x-fastest
// iteration order of meanshift.rs:114-138 for the 20^3 grid); all-zero grid -> index 0; then the caller's guesses (:437-460).
__device__ __forceinline__ void cl_initial_guess(const ClusterArgs &a, const int which, const int frame, const ClShared sh) {
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    unsigned long long *red64 = sh.red64;
    uint32_t *red32 = sh.red32;
    int32_t *s_pos = sh.s_pos;
    const uint8_t gmask = a.guess_mask ? a.guess_mask[frame] : 3;
    {
        const uint32_t *g = which == 0 ? a.pos_grid + (size_t)frame * DH_POSGRID : a.rot_grid + (size_t)frame * DH_GRID3;
        const int ncell = which == 0 ? DH_POSGRID : DH_GRID3;
        unsigned long long best = 0;   // (value << 32) | ~idx
        for (int i = tid; i < ncell; i += CL_THREADS) {
            uint32_t gv = g[i];
            unsigned long long k = ((unsigned long long)gv << 32) | (uint32_t)(~(uint32_t)i);
            if (gv && k > best) best = k;
        }
        for (int d = WAVE / 2; d; d >>= 1) { unsigned long long o = __shfl_down(best, d); if (o > best) best = o; }
        if (lane == 0) red64[wave] = best;
        __syncthreads();
        best = red64[0];
        for (int i = 1; i < CL_WAVES; ++i) if (red64[i] > best) best = red64[i];
        const uint32_t best_idx = best ? ~(uint32_t)best : 0u;
        __syncthreads();
        if (which == 0) {
            int gpw = a.w / DH_GRID, gph = a.h / DH_GRID;                 // :706-707
            int mxg = best_idx % DH_GRID, myg = best_idx / DH_GRID;       // :708-709
            // mean of the non-zero pixels of that image cell (:711-725)
            const uint16_t *img = a.frames + (size_t)frame * a.w * a.h;
            unsigned long long zs = 0; uint32_t zc = 0;
            for (int i = tid; i < gpw * gph; i += CL_THREADS) {
                int xx = gpw * mxg + i % gpw, yy = gph * myg + i / gpw;
                uint32_t v = img[(size_t)yy * a.w + xx];
                if (v) { zs += v; zc++; }
            }
            for (int d = WAVE / 2; d; d >>= 1) { zs += __shfl_down(zs, d); zc += __shfl_down(zc, d); }
            if (lane == 0) { red64[wave] = zs; red32[wave] = zc; }
            __syncthreads();
            if (tid == 0) {
                zs = 0; zc = 0;
                for (int i = 0; i < CL_WAVES; ++i) { zs += red64[i]; zc += red32[i]; }
                float meanz = zc ? (float)__ddiv_rn((double)zs, (double)zc) : 0.0f;
                float mx = __fmul_rn(__fadd_rn((float)mxg, 0.5f), (float)gpw);       // :727-728
                float my = __fmul_rn(__fadd_rn((float)myg, 0.5f), (float)gph);
                float q[3];
                to3d(a.kinv, mx, my, meanz, q);                                       // :729
                int32_t gm[3] = {f32_as_i32(q[0]), f32_as_i32(q[1]), f32_as_i32(q[2]) / DH_ZSCALEFACTOR};  // :750
                if (a.midp_guess && (gmask & 1)) {                                    // :437-441
                    const float *mg = a.midp_guess + (size_t)frame * 3;
                    gm[0] = f32_as_i32(mg[0]); gm[1] = f32_as_i32(mg[1]); gm[2] = f32_as_i32(mg[2]) / DH_ZSCALEFACTOR;
                }
                s_pos[0] = gm[0]; s_pos[1] = gm[1]; s_pos[2] = gm[2];
            }
        } else if (tid == 0) {
            uint32_t rb[3] = {best_idx % DH_GRID, (best_idx / DH_GRID) % DH_GRID, best_idx / (DH_GRID * DH_GRID)};
            for (int k = 0; k < 3; ++k) {
                double deg = __ddiv_rn(__dadd_rn(__dmul_rn((double)rb[k], 360.0), 180.0), 20.0);   // :745-747
                if (a.rot_guess && (gmask & 2))                                                   // :444-453
                    deg = __dadd_rn(__ddiv_rn(__dmul_rn(a.rot_guess[(size_t)frame * 3 + k], 180.0), 3.14159), 180.0);
                s_pos[k] = f64_as_i32(__ddiv_rn(__dmul_rn(deg, 120.0), 360.0));                   // :458-460
            }
        }
    }
}

// Adds to the (zeroed) LDS region with origin `org` every vote of accumulator `which` that falls into it, from the hit
// records [h0, h1) of the frame -- or, for rotation votes of forests with a leaf histogram, from the leaves [l0, l1).
// Integer atomics: exact and order-free, so any split of the ranges over workgroups sums to the same region.
__device__ __forceinline__ void cl_gather(const ClusterArgs &a, const int which, const int frame, const int32_t org[3],
                                          const uint32_t h0, const uint32_t h1, const uint32_t l0, const uint32_t l1, const ClShared sh) {
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    uint32_t *region = sh.region;
    float *prod = sh.prod;
    uint32_t &s_total = *sh.s_total;
    const HitRec *hits = a.hits + (size_t)frame * a.hits_cap;
    const HitBox *box = a.hit_box + (size_t)frame * a.hits_cap;
    const HitRot *hr = a.hit_rot + (size_t)frame * a.hits_cap;
    if (which == 0) {
        // Two steps: (1) every thread tests the vote bounding boxes of its records against the region and
        // appends the survivors to a list (in `prod`, idle now; a record that finds the list full is
        // handled by its thread alone); (2) 16 lanes share each listed record and take its offset votes
        // 16 apart, so the chain of dependent vote loads per lane is n_votes / 16 long instead of n_votes.
        uint32_t *list = (uint32_t *)prod;
        if (tid == 0) s_total = 0;
        __syncthreads();
        for (uint32_t i0 = h0; i0 < h1; i0 += CL_THREADS * 2) {        // (uniform trip count: the ballots need every lane)
            int4 b0[2], b1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {                                   // two records' boxes in flight (64-VGPR budget)
                const uint32_t i = min(i0 + j * CL_THREADS + tid, h1 - 1);
                b0[j] = ((const int4 *)(box + i))[0]; b1[j] = ((const int4 *)(box + i))[1];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t i = i0 + j * CL_THREADS + tid, fc = (uint32_t)b1[j].w;
                const bool keep = i < h1 && (fc & LF_OFF) && range_hits_span(b0[j].x, b0[j].w, org[0], RG) &&
                                  range_hits_span(b0[j].y, b1[j].x, org[1], RG) && range_hits_span(b0[j].z, b1[j].y, org[2], RG);
                const unsigned long long bal = __ballot(keep);             // one LDS atomic per wave, ranks from the ballot
                uint32_t wb = 0;
                if (lane == 0 && bal) wb = atomicAdd(&s_total, (uint32_t)__popcll(bal));
                wb = __shfl(wb, 0);
                if (keep) {
                    const uint32_t slot = wb + (uint32_t)__popcll(bal & lanemask_lt());
                    if (slot < CL_LIST) list[slot] = i;
                    else cluster_add_votes(a, region, hits, box, i, 0u, 1u, org);   // list full: this thread takes the record alone
                }
            }
        }
        __syncthreads();
        const uint32_t np = min(s_total, (uint32_t)CL_LIST);
        for (uint32_t k = tid; k < np * 16u; k += CL_THREADS) cluster_add_votes(a, region, hits, box, list[k >> 4], k & 15u, 16u, org);
        __syncthreads();
    } else if (a.leaf_hits) {
        // Rotation votes depend only on the leaf (prediction.rs:601-636): the accumulator is
        // sum over leaves of (times the leaf voted) x (its distinct cells), so the gather walks
        // the leaves that voted at all instead of every hit -- u32 wrap-around makes
        // hits * v * mult the same residue as that many separate adds.
        const uint32_t *lh = a.leaf_hits + (size_t)frame * a.f.n_leaves;
        uint32_t *list = (uint32_t *)prod;                    // same two-step scheme as the position gather
        for (uint32_t c0 = l0; c0 < l1; c0 += CL_LIST) {
            if (tid == 0) s_total = 0;
            __syncthreads();
            const uint32_t c1 = min(l1, c0 + CL_LIST);
            for (uint32_t l0 = c0; l0 < c1; l0 += CL_THREADS) {          // (uniform trip count: the ballot needs every lane)
                const uint32_t l = l0 + tid;
                bool keep = false;
                if (l < c1 && lh[l]) {
                    const uint4 t2 = ((const uint4 *)(a.f.tpl + l))[2];
                    const uint32_t bl = t2.y, bh = t2.z;
                    keep = bl != 0xFFFFFFFFu && range_hits_span((int32_t)(bl & 255u), (int32_t)(bh & 255u), org[0], RG) &&
                           range_hits_span((int32_t)((bl >> 8) & 255u), (int32_t)((bh >> 8) & 255u), org[1], RG) &&
                           range_hits_span((int32_t)((bl >> 16) & 255u), (int32_t)((bh >> 16) & 255u), org[2], RG);
                }
                const unsigned long long bal = __ballot(keep);             // one LDS atomic per wave, ranks from the ballot
                uint32_t wb = 0;
                if (lane == 0 && bal) wb = atomicAdd(&s_total, (uint32_t)__popcll(bal));
                wb = __shfl(wb, 0);
                if (keep) list[wb + (uint32_t)__popcll(bal & lanemask_lt())] = l;
            }
            __syncthreads();
            const uint32_t np = s_total;
            for (uint32_t k = tid; k < np * 16u; k += CL_THREADS) {
                const uint32_t l = list[k >> 4], sub = k & 15u;
                const uint4 *tp = (const uint4 *)(a.f.tpl + l);
                const uint4 t1 = tp[1], t2 = tp[2], t3 = tp[3];
                const uint32_t v = lh[l] * t1.z, q1 = t2.w + (t3.x & 0xffffu);   // times the leaf voted x valtoadd
                for (uint32_t q = t2.w + sub; q < q1; q += 16u) {
                    const uint32_t b = a.f.rot_bin[q], vm = v * a.f.rot_mult[q];                                  // prediction.rs:635
                    uint32_t dx = (b & 255u) - (uint32_t)org[0];
                    uint32_t dy = ((b >> 8) & 255u) - (uint32_t)org[1];
                    uint32_t dz = ((b >> 16) & 255u) - (uint32_t)org[2];
                    if (dx < RG && dy < RG && dz < RG) atomicAdd(&region[(dx * RG + dy) * RG + dz], vm);
                }
            }
            __syncthreads();
        }
    } else {
        for (uint32_t i0 = h0; i0 < h1; i0 += CL_THREADS * 2) {
            uint4 r[2];
            uint32_t vv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t i = i0 + j * CL_THREADS + tid;
                r[j].x = 0xFFFFFFFFu;
                if (i < h1) { r[j] = *(const uint4 *)(hr + i); vv[j] = box[i].v; }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t bl = r[j].x, bh = r[j].y;
                if (bl == 0xFFFFFFFFu) continue;
                if (!range_hits_span((int32_t)(bl & 255u), (int32_t)(bh & 255u), org[0], RG)) continue;
                if (!range_hits_span((int32_t)((bl >> 8) & 255u), (int32_t)((bh >> 8) & 255u), org[1], RG)) continue;
                if (!range_hits_span((int32_t)((bl >> 16) & 255u), (int32_t)((bh >> 16) & 255u), org[2], RG)) continue;
                const uint32_t v = vv[j];
                for (uint32_t q = r[j].z; q < r[j].z + (r[j].w & 0xffffu); ++q) {
                    const uint32_t b = a.f.rot_bin[q], vm = v * a.f.rot_mult[q];                                   // prediction.rs:635
                    uint32_t dx = (b & 255u) - (uint32_t)org[0];
                    uint32_t dy = ((b >> 8) & 255u) - (uint32_t)org[1];
                    uint32_t dz = ((b >> 16) & 255u) - (uint32_t)org[2];
                    if (dx < RG && dy < RG && dz < RG) atomicAdd(&region[(dx * RG + dy) * RG + dz], vm);
                }
            }
        }
    }
}

__global__ void __launch_bounds__(CL_THREADS, 8) k_cluster(ClusterArgs a) {
    __shared__ uint32_t region[RG3];
    __shared__ __attribute__((aligned(16))) float prod[CL_PROD_CAP * 4];
    __shared__ uint32_t cnt[CL_CHUNKS * CL_WAVES];
    __shared__ unsigned long long red64[CL_WAVES];
    __shared__ uint32_t red32[CL_WAVES];
    __shared__ int32_t s_pos[3];
    __shared__ float s_acc[4];
    __shared__ uint32_t s_total;

    const int which = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;

    const ClShared sh{region, prod, red64, red32, s_pos, &s_total};
    cl_initial_guess(a, which, frame, sh);
    __syncthreads();
    int32_t pos[3] = {s_pos[0], s_pos[1], s_pos[2]};
    if (a.dbg_guess && tid < 3) a.dbg_guess[(size_t)frame * 6 + which * 3 + tid] = pos[tid];
    int32_t *trace = a.dbg_trace ? a.dbg_trace + ((size_t)which * a.n_frames + frame) * (a.iterations + 1) * 3 : nullptr;
    if (trace && tid < 3) trace[tid] = pos[tid];

    if (KNOB_STOP(a.stop == 1)) return;
    // ---------------- mean shift (meanshift.rs:328-407)
    uint32_t n_hits = a.hit_count[frame];
    if (n_hits > a.hits_cap) n_hits = a.hits_cap;
    int32_t org[3] = {0, 0, 0};     // region origin (cell coordinates of region[0])
    bool have_region = false;
    uint32_t steps = 0;
    for (uint32_t it = 0; it < a.iterations; ++it) {
        // window offset inside the region; the region is valid while 0 <= woff <= RG-20 on every axis
        uint32_t wo0 = (uint32_t)pos[0] - 10u - (uint32_t)org[0], wo1 = (uint32_t)pos[1] - 10u - (uint32_t)org[1],
                 wo2 = (uint32_t)pos[2] - 10u - (uint32_t)org[2];
        if (!have_region || wo0 > RG - 20 || wo1 > RG - 20 || wo2 > RG - 20) {
            // ---- (re)build the region centred on the window
            for (int k = 0; k < 3; ++k) org[k] = (int32_t)((uint32_t)pos[k] - 10u - (uint32_t)((RG - 20) / 2));
            wo0 = wo1 = wo2 = (RG - 20) / 2;
            have_region = true;
            __syncthreads();                       // previous iteration's readers are done
            if (a.pre_region && it == 0 && n_hits >= a.pre_min_hits) {
                // the region around the initial guess was gathered by k_region (several workgroups per frame)
                const uint32_t *pr = a.pre_region + ((size_t)frame * 2 + which) * RG3;
                for (int i = tid; i < RG3; i += CL_THREADS) region[i] = pr[i];
            } else {
                for (int i = tid; i < RG3; i += CL_THREADS) region[i] = 0;
                __syncthreads();
                cl_gather(a, which, frame, org, 0u, n_hits, 0u, a.f.n_leaves, sh);
            }
        }
        __syncthreads();
        if (KNOB_STOP(a.stop == 2)) return;
        // ---- order-preserving compaction of the window's non-zero cells; window cell index
        // (dx*20+dy)*20+dz = chunk*1024 + tid is the reference's summation order
#pragma unroll 1
        for (int c = 0; c < CL_CHUNKS; ++c) {
            const uint32_t cell = (uint32_t)(c * CL_THREADS + tid);
            uint32_t fv = 0;
            if (cell < DH_GRID3) {
                uint32_t dz = cell % 20u, dy = (cell / 20u) % 20u, dx = cell / 400u;
                fv = region[((wo0 + dx) * RG + wo1 + dy) * RG + wo2 + dz];
            }
            uint64_t b = __ballot(fv != 0);
            if (lane == 0) cnt[c * CL_WAVES + wave] = (uint32_t)__popcll(b);
        }
        __syncthreads();
        if (wave == 0) {   // exclusive scan of the 128 (chunk, wave) counts, 2 per lane
            uint32_t c0 = cnt[lane * 2], c1 = cnt[lane * 2 + 1];
            uint32_t incl = c0 + c1;
            for (int d = 1; d < WAVE; d <<= 1) { uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            uint32_t ex = incl - (c0 + c1);
            cnt[lane * 2] = ex; cnt[lane * 2 + 1] = ex + c0;
            if (lane == WAVE - 1) s_total = incl;
            if (lane < 4) s_acc[lane] = 0.0f;
        }
        __syncthreads();
        const uint32_t total = s_total;
        for (uint32_t base = 0; base < total; base += CL_PROD_CAP) {
#pragma unroll 1
            for (int c = 0; c < CL_CHUNKS; ++c) {
                const uint32_t cell = (uint32_t)(c * CL_THREADS + tid);
                const uint32_t dz = cell % 20u, dy = (cell / 20u) % 20u, dx = cell / 400u;
                const uint32_t fv = cell < DH_GRID3 ? region[((wo0 + dx) * RG + wo1 + dy) * RG + wo2 + dz] : 0u;
                uint64_t b = __ballot(fv != 0);
                if (fv) {
                    uint32_t k = cnt[c * CL_WAVES + wave] + (uint32_t)__popcll(b & lanemask_lt());
                    // this sweep only parks (cell, value) in the cell's slot: no global load sits inside it
                    if (k >= base && k < base + CL_PROD_CAP) *(uint2 *)(prod + (k - base) * 4) = make_uint2(cell, fv);
                }
            }
            __syncthreads();
            {   // one thread per parked cell: Gaussian weight (one global load, all in flight together) and products, in place
                const uint32_t m = min((uint32_t)CL_PROD_CAP, total - base);
                if ((uint32_t)tid < m) {
                    const uint2 cf = *(const uint2 *)(prod + tid * 4);
                    const uint32_t cell = cf.x, dz = cell % 20u, dy = (cell / 20u) % 20u, dx = cell / 400u;
                    float w = __fmul_rn(a.kern_ord[cell], (float)cf.y);                     // meanshift.rs:370-379
                    float ax = (float)(int32_t)((uint32_t)pos[0] + dx - 10u);               // :373-375
                    float ay = (float)(int32_t)((uint32_t)pos[1] + dy - 10u);
                    float az = (float)(int32_t)((uint32_t)pos[2] + dz - 10u);
                    *(float4 *)(prod + tid * 4) = make_float4(__fmul_rn(ax, w), __fmul_rn(ay, w), __fmul_rn(az, w), w);
                }
            }
            __syncthreads();
            if (tid < 4) {   // the sequential chain: acc = acc + prod[i], in cell order
                float acc = s_acc[tid];
                uint32_t m = min((uint32_t)CL_PROD_CAP, total - base);
                uint32_t i = 0;
                for (; i + 8 <= m; i += 8) {
                    float v0 = prod[(i + 0) * 4 + tid], v1 = prod[(i + 1) * 4 + tid], v2 = prod[(i + 2) * 4 + tid],
                          v3 = prod[(i + 3) * 4 + tid], v4 = prod[(i + 4) * 4 + tid], v5 = prod[(i + 5) * 4 + tid],
                          v6 = prod[(i + 6) * 4 + tid], v7 = prod[(i + 7) * 4 + tid];
                    acc = __fadd_rn(acc, v0); acc = __fadd_rn(acc, v1); acc = __fadd_rn(acc, v2); acc = __fadd_rn(acc, v3);
                    acc = __fadd_rn(acc, v4); acc = __fadd_rn(acc, v5); acc = __fadd_rn(acc, v6); acc = __fadd_rn(acc, v7);
                }
                for (; i < m; ++i) acc = __fadd_rn(acc, prod[i * 4 + tid]);
                s_acc[tid] = acc;
            }
            __syncthreads();
        }
        if (KNOB_STOP(a.stop == 3)) return;
        const float den = s_acc[3];
        if (den == 0.0f) break;                                                              // :385-388
        int32_t np0 = f32_as_i32(__fdiv_rn(s_acc[0], den)), np1 = f32_as_i32(__fdiv_rn(s_acc[1], den)),
                np2 = f32_as_i32(__fdiv_rn(s_acc[2], den));                                   // :391-394
        const bool fixed = np0 == pos[0] && np1 == pos[1] && np2 == pos[2];
        pos[0] = np0; pos[1] = np1; pos[2] = np2;
        steps++;
        if (trace && tid < 3) trace[steps * 3 + tid] = pos[tid];
        if (fixed) {
            // a fixed point: every remaining iteration sees the same window and returns the same
            // position, so the reference's result (and trace) is this position repeated
            if (trace && tid < 3)
                for (uint32_t s2 = steps + 1; s2 <= a.iterations; ++s2) trace[s2 * 3 + tid] = pos[tid];
            steps = a.iterations;
            break;
        }
    }
    if (a.dbg_steps && tid == 0) a.dbg_steps[(size_t)which * a.n_frames + frame] = steps;
    if (tid == 0) {
        dh_pose *o = a.out + frame;
        if (which == 0) {                                                                    // prediction.rs:486-488
            o->mid_point[0] = (float)pos[0];
            o->mid_point[1] = (float)pos[1];
            o->mid_point[2] = (float)(int32_t)((uint32_t)pos[2] * (uint32_t)DH_ZSCALEFACTOR);
            o->reserved = 0;
        } else {                                                                             // :477-482
            for (int k = 0; k < 3; ++k)
                o->rotation[k] = __dmul_rn(__ddiv_rn(__dsub_rn((double)pos[k], 60.0), 60.0), 3.14159);
        }
    }
}

hipError_t dh_launch_cluster(const ClusterArgs &a, hipStream_t s) {
    if (a.n_frames == 0) return hipSuccess;
    hipLaunchKernelGGL(k_cluster, dim3(2, a.n_frames), dim3(CL_THREADS), 0, s, a);
    return hipGetLastError();
}

// ================================================================== k_region
// The first region gather of k_cluster, spread over several workgroups per (frame, accumulator): with few frames in the
// batch and many hit records per frame (large forests, stride 1-2: 60-90 k records per frame at BASELINE config 3) one
// workgroup streaming a whole frame's records is the slowest thing in the step while most CUs idle.  Workgroup
// (slice, frame, which) recomputes the initial guess (a few microseconds), gathers its share of the records (or leaves)
// into an LDS region exactly as k_cluster would and adds its non-zero cells to the frame's pre-built region in global
// memory (integer atomics: the sum over the slices is the region k_cluster would have built).
__global__ void __launch_bounds__(CL_THREADS, 8) k_region(ClusterArgs a) {
    __shared__ uint32_t region[RG3];
    __shared__ __attribute__((aligned(16))) float prod[CL_PROD_CAP * 4];
    __shared__ unsigned long long red64[CL_WAVES];
    __shared__ uint32_t red32[CL_WAVES];
    __shared__ int32_t s_pos[3];
    __shared__ uint32_t s_total;
    const int slice = blockIdx.x, frame = blockIdx.y, which = blockIdx.z, tid = threadIdx.x;
    const ClShared sh{region, prod, red64, red32, s_pos, &s_total};
    uint32_t n_hits = a.hit_count[frame];
    if (n_hits > a.hits_cap) n_hits = a.hits_cap;
    if (n_hits < a.pre_min_hits) return;                            // few records: k_cluster gathers this frame's regions itself
    // this workgroup's share: hit records in whole rounds of the gather loops, leaves in whole list chunks
    const uint32_t S = (uint32_t)a.pre_slices;
    const uint32_t hper = ((n_hits + S - 1) / S + 2 * CL_THREADS - 1) / (2 * CL_THREADS) * (2 * CL_THREADS);
    const uint32_t h0 = min(n_hits, (uint32_t)slice * hper), h1 = min(n_hits, h0 + hper);
    const uint32_t lper = ((a.f.n_leaves + S - 1) / S + CL_LIST - 1) / CL_LIST * CL_LIST;
    const uint32_t l0 = min(a.f.n_leaves, (uint32_t)slice * lper), l1 = min(a.f.n_leaves, l0 + lper);
    const bool by_leaves = which == 1 && a.leaf_hits;
    if (by_leaves ? l0 >= l1 : h0 >= h1) return;                    // nothing in this share (uniform for the workgroup)
    cl_initial_guess(a, which, frame, sh);
    __syncthreads();
    int32_t org[3];
    for (int k = 0; k < 3; ++k) org[k] = (int32_t)((uint32_t)s_pos[k] - 10u - (uint32_t)((RG - 20) / 2));
    for (int i = tid; i < RG3; i += CL_THREADS) region[i] = 0;
    __syncthreads();
    cl_gather(a, which, frame, org, h0, h1, l0, l1, sh);
    __syncthreads();
    uint32_t *pr = a.pre_region + ((size_t)frame * 2 + which) * RG3;
    for (int i = tid; i < RG3; i += CL_THREADS) {
        const uint32_t v = region[i];
        if (v) atomicAdd(&pr[i], v);
    }
}

hipError_t dh_launch_region(const ClusterArgs &a, hipStream_t s) {
    if (a.n_frames == 0 || a.iterations == 0 || !a.pre_region || a.pre_slices < 1) return hipSuccess;
    if (a.n_frames > 65535) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_region, dim3(a.pre_slices, a.n_frames, 2), dim3(CL_THREADS), 0, s, a);
    return hipGetLastError();
}

// ================================================================== k_votes_dump (parity tap)
// Emits every vote of one frame as an (x, y, z, value) record so a test can aggregate them into
// the full sparse accumulator the reference builds (prediction.rs:635, :667).
__global__ void __launch_bounds__(256) k_votes_dump(VotesDumpArgs a) {
    uint32_t n = a.hit_count[a.frame];
    if (n > a.hits_cap) n = a.hits_cap;
    const HitRec *hits = a.hits + (size_t)a.frame * a.hits_cap;
    const HitBox *box = a.hit_box + (size_t)a.frame * a.hits_cap;
    const HitRot *hr = a.hit_rot + (size_t)a.frame * a.hits_cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 rec = *(const float4 *)(hits + i);
        const uint32_t v = box[i].v, fc = box[i].fc;
        if (a.which == 0 && (fc & LF_OFF)) {
            const uint32_t ob = __float_as_uint(rec.w);
            for (uint32_t o = ob; o < ob + (fc >> 8); ++o) {
                const float4 of = a.f.off4[o];
                float nx = __fsub_rn(rec.x, of.x), ny = __fsub_rn(rec.y, of.y), nz = __fsub_rn(rec.z, of.z);
                if (nz < 0.0f) continue;
                uint32_t k = atomicAdd(a.count, 1u);
                if (k < a.cap) {
                    a.out[k * 4 + 0] = f32_as_i32(nx); a.out[k * 4 + 1] = f32_as_i32(ny);
                    a.out[k * 4 + 2] = f32_as_i32(__fdiv_rn(nz, (float)DH_ZSCALEFACTOR)); a.out[k * 4 + 3] = (int32_t)v;
                }
            }
        } else if (a.which == 1 && (fc & LF_ROT)) {
            for (uint32_t r = hr[i].rb; r < hr[i].rb + (hr[i].n_rot & 0xffffu); ++r) {
                uint32_t b = a.f.rot_bin[r];
                uint32_t k = atomicAdd(a.count, 1u);
                if (k < a.cap) {
                    a.out[k * 4 + 0] = (int32_t)(b & 255u); a.out[k * 4 + 1] = (int32_t)((b >> 8) & 255u);
                    a.out[k * 4 + 2] = (int32_t)((b >> 16) & 255u); a.out[k * 4 + 3] = (int32_t)(v * a.f.rot_mult[r]);
                }
            }
        }
    }
}

hipError_t dh_launch_votes_dump(const VotesDumpArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_votes_dump, dim3(256), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ================================================================== predict_mask / 2-D Hough votes
// HoughPrediction::predict_mask (prediction.rs:850-905): one thread per window position.
__device__ __forceinline__ uint8_t f64_as_u8(double v) {
    if (v != v || v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

__global__ void __launch_bounds__(256) k_mask(AuxArgs a) {
    const int frame = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x, npatch = a.nx * a.ny;
    if (p >= npatch) return;
    const size_t po = (size_t)frame * npatch + p;
    if (!(a.flags[po] & 1)) return;                                   // background window (:870-878)
    const int T = (int)a.f.n_trees;
    double prob = 0.0;                                                // :881-882, tree order
    for (int t = 0; t < T; ++t) prob = __dadd_rn(prob, a.f.leaf_prob[a.leaf[po * T + t]]);
    prob = __ddiv_rn(prob, (double)T);
    const uint8_t pv = f64_as_u8(__dmul_rn(prob, 255.0));             // :883
    const uint32_t x = (uint32_t)(a.lw + (p % a.nx) * a.step), y = (uint32_t)(a.lh + (p / a.nx) * a.step);
    const uint32_t step = (uint32_t)a.step, half = step / 2;
    uint8_t *m = a.mask + (size_t)frame * a.w * a.h;
    for (uint32_t i = 0; i < step; ++i)                               // :884-897
        for (uint32_t j = 0; j < step; ++j) {
            if (x + i < half || y + j < half) continue;
            if (x + i - half >= (uint32_t)a.w || y + j - half >= (uint32_t)a.h) continue;
            m[(size_t)(y + j - half) * a.w + (x + i - half)] = pv;
        }
}

// Voting stage of HoughPrediction::build_hough_image (prediction.rs:760-840): one thread per
// (window position, tree).  The u16 image of the reference wraps modulo 2^16; votes are summed in
// 32 bits with integer atomics and narrowed afterwards, which is the same residue.
__global__ void __launch_bounds__(256) k_hough2d(AuxArgs a) {
    const int frame = blockIdx.y, T = (int)a.f.n_trees, npatch = a.nx * a.ny;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npatch * T) return;
    const int p = i / T, t = i - p * T;
    const size_t po = (size_t)frame * npatch + p;
    if (!(a.flags[po] & 1)) return;                                   // :790-798
    const uint32_t L = (uint32_t)a.leaf[po * T + t];
    const double lp = a.f.leaf_prob[L];
    if (!(lp >= 0.95)) return;                                        // :805
    const uint32_t ob = a.f.off_begin[L], oe = a.f.off_begin[L + 1];
    if (oe == ob) return;                                             // excluded by forest validation
    const uint32_t val = (uint32_t)(uint16_t)(f64_as_usize(__dmul_rn(255.0, lp)) / (uint64_t)(oe - ob));   // :807-808
    const int x = a.lw + (p % a.nx) * a.step, y = a.lh + (p / a.nx) * a.step;
    const uint16_t *img = a.frames + (size_t)frame * a.w * a.h;
    float p3[3];
    to3d(a.kinv, (float)x, (float)y, (float)img[(size_t)y * a.w + x], p3);          // :777-779
    uint32_t *out = a.hough32 + (size_t)frame * a.w * a.h;
    for (uint32_t o = ob; o < oe; ++o) {                              // :813
        const float *of = a.f.offsets + (size_t)o * 3;
        float r[3];
        matvec3(a.k, __fsub_rn(p3[0], of[0]), __fsub_rn(p3[1], of[1]), __fsub_rn(p3[2], of[2]), r);   // :814-815
        const int32_t vx = f32_as_i32(__fdiv_rn(r[0], r[2])), vy = f32_as_i32(__fdiv_rn(r[1], r[2]));   // :816
        if (vx < 0 || vx >= a.w || vy < 0 || vy >= a.h) continue;     // :818-831
        atomicAdd(&out[(size_t)vy * a.w + vx], val);                  // :832
    }
}

__global__ void __launch_bounds__(256) k_narrow_u16(const uint32_t *in, uint16_t *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint16_t)in[i];
}

hipError_t dh_launch_mask(const AuxArgs &a, hipStream_t s) {
    const int npatch = a.nx * a.ny;
    if (a.n_frames == 0 || npatch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mask, dim3((npatch + 255) / 256, a.n_frames), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t dh_launch_hough2d(const AuxArgs &a, uint16_t *out, hipStream_t s) {
    const int pairs = a.nx * a.ny * (int)a.f.n_trees;
    if (a.n_frames == 0) return hipSuccess;
    if (pairs > 0) hipLaunchKernelGGL(k_hough2d, dim3((pairs + 255) / 256, a.n_frames), dim3(256), 0, s, a);
    const size_t n = (size_t)a.n_frames * a.w * a.h;
    hipLaunchKernelGGL(k_narrow_u16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.hough32, out, n);
    return hipGetLastError();
}

// ================================================================== k_rle_decode
// Device side of read_depth (/root/reference src/db_reader/biwi.rs:81-103): the run-length coded depth payloads of a
// frame batch have been uploaded as they are (one blob, every frame's bytes at a 16-byte aligned offset) together with
// a run table the host built while validating the headers: per non-empty run its first destination pixel (index in
// the batch's frame array) and the position of its first depth value in the blob (in u16 units; the run's length is
// the u32 right in front of it, :94).  The frames are zero-filled by a memset (the empty runs, :90-93); here every
// wave copies runs: lane i moves values i, i + 64, ... (:95-98).  Byte / index work only.
#define RLE_THREADS 256
__global__ void __launch_bounds__(RLE_THREADS) k_rle_decode(RleArgs a) {
    const int frame = blockIdx.y, lane = threadIdx.x & (WAVE - 1);
    const uint32_t r0 = a.run_begin[frame], r1 = a.run_begin[frame + 1];
    const uint32_t wave = blockIdx.x * (RLE_THREADS / WAVE) + (threadIdx.x >> 6), nwaves = gridDim.x * (RLE_THREADS / WAVE);
    for (uint32_t r = r0 + wave; r < r1; r += nwaves) {
        const uint2 e = a.runs[r];                            // dst pixel, src u16 index
        const uint32_t n_full = (uint32_t)a.blob[e.y - 2] | ((uint32_t)a.blob[e.y - 1] << 16);
        const uint16_t *src = a.blob + e.y;
        uint16_t *dst = a.frames + e.x;
        for (uint32_t i = lane; i < n_full; i += WAVE) dst[i] = src[i];
    }
}

hipError_t dh_launch_rle_decode(const RleArgs &a, hipStream_t s) {
    if (a.n_frames == 0) return hipSuccess;
    if (a.n_frames > 65535) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_rle_decode, dim3(a.blocks_per_frame, a.n_frames), dim3(RLE_THREADS), 0, s, a);
    return hipGetLastError();
}

// ================================================================== 2-D Hough variant: blur + argmax
// imageproc::filter::gaussian_blur_f32 (imageproc 0.12.0, /root/reference Cargo.lock:555; called at
// src/hough/prediction.rs:844) = separable_filter_equal(image, gaussian_kernel_f32(sigma)): a horizontal pass
// writing a u16 image, then a vertical pass over that image.  Per output pixel: acc = 0; for every tap i in kernel
// order acc = acc + (f32)pixel[clamped position] * kernel[i] (f32, multiply then add); result = Clamp<f32> for u16
// (x >= 65535 -> 65535, x <= 0 -> 0, else truncation).  Image borders replicate the edge pixel.
// PARITY UNPINNED: the crate's source is not in the container; this restates its published algorithm.
#define BLUR_TILE 256
template <bool VERT>
__global__ void __launch_bounds__(BLUR_TILE) k_blur_u16(const uint16_t *in, uint16_t *out, int w, int h, const float *kern, int klen) {
    const int frame = blockIdx.z;
    const size_t fo = (size_t)frame * w * h;
    const int x = VERT ? (int)(blockIdx.x * BLUR_TILE + threadIdx.x) : (int)(blockIdx.x * BLUR_TILE + threadIdx.x);
    const int y = (int)blockIdx.y;
    if (x >= w) return;
    const int half = klen / 2;
    float acc = 0.0f;
    if (VERT) {
        for (int i = 0; i < klen; ++i) {
            const int yy = min(max(y + i - half, 0), h - 1);
            acc = __fadd_rn(acc, __fmul_rn((float)in[fo + (size_t)yy * w + x], kern[i]));
        }
    } else {
        const uint16_t *row = in + fo + (size_t)y * w;
        for (int i = 0; i < klen; ++i) {
            const int xx = min(max(x + i - half, 0), w - 1);
            acc = __fadd_rn(acc, __fmul_rn((float)row[xx], kern[i]));
        }
    }
    uint16_t r;
    if (acc < 65535.0f) r = acc > 0.0f ? (uint16_t)acc : (uint16_t)0;      // Clamp<f32> for u16; NaN -> 65535 like the crate's comparison chain
    else r = 65535;
    out[fo + (size_t)y * w + x] = r;
}

hipError_t dh_launch_blur_u16(const uint16_t *in, uint16_t *tmp, uint16_t *out, int n, int w, int h, const float *kern, int klen, hipStream_t s) {
    if (n == 0 || w == 0 || h == 0) return hipSuccess;
    if (h > 65535 || n > 65535) return hipErrorInvalidConfiguration;
    const dim3 grid((w + BLUR_TILE - 1) / BLUR_TILE, h, n);
    hipLaunchKernelGGL(k_blur_u16<false>, grid, dim3(BLUR_TILE), 0, s, in, tmp, w, h, kern, klen);
    hipLaunchKernelGGL(k_blur_u16<true>, grid, dim3(BLUR_TILE), 0, s, (const uint16_t *)tmp, out, w, h, kern, klen);
    return hipGetLastError();
}

// HoughPrediction::predict_parameter_from2dhough (prediction.rs:343-367): `max_by_key` over the pixel indices returns
// the LAST index holding the greatest value; the head position is that pixel lifted with the frame's depth there.
__global__ void __launch_bounds__(1024) k_argmax2d(const uint16_t *hough, const uint16_t *frames, int w, int h, Mat3Arg kinv, dh_pose *out) {
    __shared__ unsigned long long red[16];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const size_t fo = (size_t)frame * w * h;
    const uint32_t npx = (uint32_t)w * (uint32_t)h;
    unsigned long long best = 0;                                           // (value << 32) | index: greatest value, then greatest index
    for (uint32_t i = tid; i < npx; i += 1024) {
        const unsigned long long k = ((unsigned long long)hough[fo + i] << 32) | i;
        if (k >= best) best = k;
    }
    for (int d = WAVE / 2; d; d >>= 1) { const unsigned long long o = __shfl_down(best, d); if (o > best) best = o; }
    if ((tid & (WAVE - 1)) == 0) red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 16; ++i) if (red[i] > best) best = red[i];
        const uint32_t idx = (uint32_t)best;
        const uint32_t x = idx % (uint32_t)w, y = idx / (uint32_t)w;       // :357-358
        const uint16_t z = frames[fo + idx];                               // :359
        float p[3];
        to3d(kinv.m, (float)x, (float)y, (float)z, p);                     // :360
        dh_pose r;
        r.mid_point[0] = p[0]; r.mid_point[1] = p[1]; r.mid_point[2] = p[2];
        r.reserved = 0;
        r.rotation[0] = 0.0; r.rotation[1] = 0.0; r.rotation[2] = 0.0;     // :363
        out[frame] = r;
    }
}

hipError_t dh_launch_argmax2d(const uint16_t *hough, const uint16_t *frames, int n, int w, int h, const float kinv[9], dh_pose *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    Mat3Arg k;
    for (int i = 0; i < 9; ++i) k.m[i] = kinv[i];
    hipLaunchKernelGGL(k_argmax2d, dim3(n), dim3(1024), 0, s, hough, frames, w, h, k, out);
    return hipGetLastError();
}

// ================================================================== k_zero
// Zero-fill of the per-batch counters inside a captured hipGraph: a memset node of tens of megabytes was observed to
// leave part of the range untouched on replay (ROCm 7.2; tools/soak.py found poses going wrong from the second replay of a
// graph captured on a 512-frame workspace), a kernel node does what it says.
__global__ void __launch_bounds__(256) k_zero(uint4 *p, size_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
}

hipError_t dh_launch_zero(void *ptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((((size_t)ptr) | bytes) & 15) return hipErrorInvalidValue;
    const size_t n16 = bytes / 16;
    const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 4096);
    hipLaunchKernelGGL(k_zero, dim3(blocks), dim3(256), 0, s, (uint4 *)ptr, n16);
    return hipGetLastError();
}
