// dh_host.cpp -- host-only logic of the runtime (see dh_host.h).  Plain C++17: builds with g++ (sanitizer builds of
// tests/host/) and with hipcc (the product library).  Compiled -ffp-contract=off: the f32 tables keep the reference's
// separate multiply / add rounding.
#include "dh_host.h"

#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <string>
#include <thread>
#include <utility>

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

int dh_fail_(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
const char *dh_err_get_(void) { return g_err; }
void dh_err_set_(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

// ------------------------------------------------------------------ forest
static inline int32_t rot_bin_host(double deg) {
    // (deg * 120 / 360) as i32 + 60 (prediction.rs:605-613); only used to validate the forest
    double v = deg * 120.0 / 360.0;
    int32_t r;
    if (v != v) r = 0;
    else if (v >= 2147483648.0) r = INT32_MAX;
    else if (v <= -2147483648.0) r = INT32_MIN;
    else r = (int32_t)v;
    return (int32_t)((uint32_t)r + 60u);
}

int dh_forest_build_(const dh_forest_desc *d, dh_forest **out) {
    if (!d || !out) return dh_fail_(DH_EINVAL, "dh_forest_create: NULL argument");
    *out = nullptr;
    if (d->n_trees == 0 || !d->roots) return dh_fail_(DH_EINVAL, "forest has no trees");
    if (d->n_leaves == 0 || !d->leaf_prob || !d->off_begin || !d->rot_begin) return dh_fail_(DH_EINVAL, "forest has no leaves");
    if (d->n_nodes && !d->nodes) return dh_fail_(DH_EINVAL, "nodes is NULL");
    if (d->n_nodes > 0x7fffffffu || d->n_leaves > 0x7fffffffu) return dh_fail_(DH_EINVAL, "forest too large");
    const uint32_t NL = d->n_leaves, NN = d->n_nodes;
    if (d->off_begin[0] != 0 || d->rot_begin[0] != 0) return dh_fail_(DH_EFOREST, "CSR arrays must start at 0");
    for (uint32_t i = 0; i < NL; ++i)
        if (d->off_begin[i + 1] < d->off_begin[i] || d->rot_begin[i + 1] < d->rot_begin[i])
            return dh_fail_(DH_EFOREST, "CSR arrays not monotone at leaf %u", i);
    const uint32_t n_off = d->off_begin[NL], n_rot = d->rot_begin[NL];
    if ((n_off && !d->offsets) || (n_rot && !d->rotations)) return dh_fail_(DH_EINVAL, "vote arrays are NULL");

    struct Owner {   // frees the half-built forest on every early return (and when a vector throws)
        dh_forest *f;
        ~Owner() { delete f; }
    } own{new dh_forest};
    dh_forest *f = own.f;
    f->roots.assign(d->roots, d->roots + d->n_trees);
    if (NN) f->nodes.assign(d->nodes, d->nodes + NN);
    f->leaf_prob.assign(d->leaf_prob, d->leaf_prob + NL);
    f->off_begin.assign(d->off_begin, d->off_begin + NL + 1);
    f->rot_begin.assign(d->rot_begin, d->rot_begin + NL + 1);
    if (n_off) f->offsets.assign(d->offsets, d->offsets + (size_t)n_off * 3);
    if (n_rot) f->rotations.assign(d->rotations, d->rotations + (size_t)n_rot * 3);

    // ---- structure: every child in range, every node reached at most once (a forest of trees:
    // guarantees each walk ends after at most max_depth steps), rectangles well-formed
    std::vector<uint8_t> seen(NN, 0);
    std::vector<std::pair<int32_t, uint32_t>> stack;
    for (uint32_t t = 0; t < d->n_trees; ++t) {
        int32_t r = f->roots[t];
        if (r >= 0 ? (uint32_t)r >= NN : (uint32_t)(~r) >= NL) return dh_fail_(DH_EFOREST, "root of tree %u out of range", t);
        if (r < 0) continue;
        stack.clear();
        stack.push_back({r, 1u});
        while (!stack.empty()) {
            const int32_t n = stack.back().first;
            const uint32_t depth = stack.back().second;
            stack.pop_back();
            if (seen[n]) return dh_fail_(DH_EFOREST, "node %d is reachable twice (cycle or shared subtree)", n);
            seen[n] = 1;
            f->max_depth = std::max(f->max_depth, depth);
            const dh_node &nd = f->nodes[n];
            for (const uint16_t *rc : {nd.r1, nd.r2}) {
                if (rc[2] < rc[0] || rc[3] < rc[1]) return dh_fail_(DH_EFOREST, "node %d: rectangle with negative extent", n);
                f->max_x = std::max(f->max_x, rc[2]);
                f->max_y = std::max(f->max_y, rc[3]);
            }
            if (nd.threshold != nd.threshold) return dh_fail_(DH_EFOREST, "node %d: NaN threshold", n);
            for (int32_t c : {nd.child_zero, nd.child_one}) {
                if (c >= 0) {
                    if ((uint32_t)c >= NN) return dh_fail_(DH_EFOREST, "node %d: child out of range", n);
                    stack.push_back({c, depth + 1});
                } else if ((uint32_t)(~c) >= NL) {
                    return dh_fail_(DH_EFOREST, "node %d: leaf out of range", n);
                }
            }
        }
    }
    for (uint32_t L = 0; L < NL; ++L)
        if (f->off_begin[L + 1] - f->off_begin[L] >= (1u << 24) || f->rot_begin[L + 1] - f->rot_begin[L] >= (1u << 16))
            return dh_fail_(DH_EFOREST, "leaf %u: more than 2^24 offset or 2^16 rotation votes", L);
    // ---- one rectangle size for the whole forest? (the in-tree trainer's geometry)
    if (NN > 0) {
        const dh_node &n0 = f->nodes[0];
        f->rw = (uint16_t)(n0.r1[2] - n0.r1[0]); f->rh = (uint16_t)(n0.r1[3] - n0.r1[1]);
        f->uniform = f->rw > 0 && f->rh > 0;
        for (uint32_t i = 0; i < NN && f->uniform; ++i)
            for (const uint16_t *rc : {f->nodes[i].r1, f->nodes[i].r2})
                if (rc[2] - rc[0] != f->rw || rc[3] - rc[1] != f->rh) f->uniform = false;
    }
    // ---- leaves that can vote
    for (uint32_t L = 0; L < NL; ++L) {
        if (!(f->leaf_prob[L] > 0.0)) continue;
        if (f->off_begin[L + 1] == f->off_begin[L]) return dh_fail_(DH_EFOREST, "leaf %u: prob > 0 but no offsets (reference divides by zero)", L);
        if (f->rot_begin[L + 1] == f->rot_begin[L]) return dh_fail_(DH_EFOREST, "leaf %u: prob > 0 but no rotations (reference unwraps None)", L);
        for (uint32_t i = f->rot_begin[L]; i < f->rot_begin[L + 1]; ++i)
            for (int k = 0; k < 3; ++k) {
                int32_t r = rot_bin_host(f->rotations[(size_t)i * 3 + k]);
                if (r >= DH_ROT_GRID_PARTS) r -= DH_ROT_GRID_PARTS;
                else if (r < 0) r += DH_ROT_GRID_PARTS;
                if (r < 0 || r >= DH_ROT_GRID_PARTS) return dh_fail_(DH_EFOREST, "leaf %u: rotation bin outside [0,120) after one wrap (reference indexes out of bounds)", L);
            }
    }
    own.f = nullptr;
    *out = f;
    return DH_OK;
}

void dh_pack_off4_(const dh_forest &f, std::vector<uint32_t> &b4, std::vector<float> &o4) {
    const uint32_t NL = (uint32_t)f.leaf_prob.size();
    b4.assign((size_t)NL + 1, 0);
    for (uint32_t L = 0; L < NL; ++L) b4[L + 1] = b4[L] + ((f.off_begin[L + 1] - f.off_begin[L] + 3u) & ~3u);
    o4.assign(((size_t)b4[NL] + 4) * 4, 0.0f);
    for (uint32_t L = 0; L < NL; ++L)
        for (uint32_t k = f.off_begin[L]; k < f.off_begin[L + 1]; ++k) {
            float *o = &o4[((size_t)b4[L] + (k - f.off_begin[L])) * 4];
            o[0] = f.offsets[(size_t)k * 3]; o[1] = f.offsets[(size_t)k * 3 + 1]; o[2] = f.offsets[(size_t)k * 3 + 2]; o[3] = 0.0f;
        }
}

// pad covers the rounding of the two f64 products (C1 C2 < 2^32, |thr| <= 65535: the products are below 2^48, their
// rounding error below 2^-4) with room to spare.
void dh_build_nodes_g_(const dh_forest &f, std::vector<NodeG> &ng) {
    ng.resize(f.nodes.size());
    for (size_t i = 0; i < f.nodes.size(); ++i) {
        const dh_node &nd = f.nodes[i];
        NodeG o{};
        for (int k = 0; k < 4; ++k) { o.r1[k] = (uint8_t)nd.r1[k]; o.r2[k] = (uint8_t)nd.r2[k]; }
        const uint32_t c1 = (uint32_t)(nd.r1[2] - nd.r1[0]) * (uint32_t)(nd.r1[3] - nd.r1[1]);
        const uint32_t c2 = (uint32_t)(nd.r2[2] - nd.r2[0]) * (uint32_t)(nd.r2[3] - nd.r2[1]);
        const uint32_t C1 = std::max(c1, 1u), C2 = std::max(c2, 1u);            // <= 255 * 255
        o.cc = C1 | (C2 << 16);
        o.child_zero = nd.child_zero; o.child_one = nd.child_one;
        const double thr = nd.threshold, cc = (double)C1 * (double)C2;
        if (thr >= 65535.0) { o.ilo = INT64_MAX - 16; o.amb = 0; }              // mean difference <= 65535: never greater
        else if (thr < -65535.0) { o.ilo = INT64_MIN; o.amb = 0; }             // >= -65535: always greater
        else {
            const double m = 5.820766091346741e-11;                             // 2^-34
            const double lo = floor((thr - m) * cc - 1.0), hi = ceil((thr + m) * cc + 1.0);
            o.ilo = (int64_t)lo;
            o.amb = (uint32_t)std::min<int64_t>((int64_t)hi - (int64_t)lo - 1, 0xffffffffll);
        }
        ng[i] = o;
    }
}

// ------------------------------------------------------------------ diagnostic switches
Knobs dh_read_knobs_() {
    Knobs k;
    auto geti = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    k.force_general = getenv("DH_FORCE_GENERAL") != nullptr;
    k.no_leaf_hist = getenv("DH_NO_LEAF_HIST") != nullptr;
    k.box_no_ring = getenv("DH_BOX_NO_RING") != nullptr;
    k.leaf_hist_max = (uint32_t)std::max(0, geti("DH_LEAF_HIST_MAX", 16384));
    k.lds_budget_kb = std::max(0, geti("DH_LDS_BUDGET_KB", 0));
    if (const char *e = getenv("DH_TILE")) {
        if (sscanf(e, "%d,%d", &k.tile_x, &k.tile_y) != 2 || k.tile_x < 1 || k.tile_y < 1) k.tile_x = k.tile_y = 0;
    }
    k.box_band = std::max(1, geti("DH_BOX_BAND", 64));
    k.box_bands = std::max(0, geti("DH_BOX_BANDS", 0));
    k.max_resident = std::max(1, geti("DH_MAX_RESIDENT_FRAMES", 512));
    k.chunks = std::max(0, std::min(8, geti("DH_CHUNKS", 0)));
    k.box_dense = getenv("DH_BOX_DENSE") != nullptr;
    k.no_region = getenv("DH_NO_REGION") != nullptr;
    k.region_min_hits = std::max(0, geti("DH_REGION_MIN_HITS", 0));
    k.no_general_int = getenv("DH_NO_GENERAL_INT") != nullptr;
    k.no_absorb = getenv("DH_NO_ABSORB") != nullptr;
    k.vote_exact = getenv("DH_VOTE_EXACT") != nullptr;
    k.no_tile_list = getenv("DH_NO_TILE_LIST") != nullptr;
    k.no_zero_fold = getenv("DH_NO_ZERO_FOLD") != nullptr;
    if (const char *e = getenv("DH_TOP_LEVELS")) k.top_levels = std::max(0, std::min(8, atoi(e)));
    k.stage_chunk = std::max(1, geti("DH_STAGE_CHUNK", 64));
    k.host_threads = std::max(1, std::min(64, geti("DH_HOST_THREADS", (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency())))));
#ifdef DH_PROFILING_KNOBS
    k.trav_stop = geti("DH_TRAV_STOP", 0); k.emit_stop = geti("DH_EMIT_STOP", 0);
    k.vote_stop = geti("DH_VOTE_STOP", 0); k.cl_stop = geti("DH_CL_STOP", 0);
    k.trav_stamps = getenv("DH_TRAV_STAMPS") != nullptr;
    k.cl_stamps = getenv("DH_CL_STAMPS") != nullptr;
#endif
    return k;
}

// ------------------------------------------------------------------ geometry
int dh_patch_grid_(const dh_params &p, int w, int h, int *nx, int *ny) {
    if (p.stepwidth == 0 || p.subimage_width == 0 || p.subimage_height == 0) return dh_fail_(DH_EINVAL, "zero stepwidth / patch size");
    if (w <= 0 || h <= 0) return dh_fail_(DH_EINVAL, "non-positive frame size");
    // the reference computes `h - right_h` in u32 and panics in SubImage::new when the frame is
    // smaller than the patch (prediction.rs:546-548, :565)
    if ((uint32_t)w < p.subimage_width || (uint32_t)h < p.subimage_height) return dh_fail_(DH_ESIZE, "frame %dx%d smaller than the %ux%u patch", w, h, p.subimage_width, p.subimage_height);
    uint32_t lw = p.subimage_width / 2, rw = p.subimage_width - lw, lh = p.subimage_height / 2, rh = p.subimage_height - lh;
    uint32_t xe = (uint32_t)w - rw, ye = (uint32_t)h - rh;
    *nx = xe > lw ? (int)((xe - lw + p.stepwidth - 1) / p.stepwidth) : 0;
    *ny = ye > lh ? (int)((ye - lh + p.stepwidth - 1) / p.stepwidth) : 0;
    return DH_OK;
}

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
        for (int i = 0; i < 64; ++i) cnt[(unsigned)((i / px) * step * ss + (i % px) * (step / m)) & 63u]++;
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

// Tile of PX x PY window positions per workgroup: as many positions as fit the LDS budget
// (SAT footprint or box-sum region + leaf ids), at most 1024 (one thread per position in the tail).
int dh_choose_tile_(const TileQuery &p, Geom &g) {
    const int step = (int)p.params.stepwidth, sw = (int)p.params.subimage_width, sh = (int)p.params.subimage_height;
    const int rw = g.uniform ? p.f_rw : 0, rh = g.uniform ? p.f_rh : 0;
    size_t budget = 79 * 1024;   // two 1024-thread workgroups per CU (160 KB LDS)
    if (p.lds_budget_kb > 0) budget = (size_t)p.lds_budget_kb * 1024;
    budget = std::min<size_t>(budget, 158 * 1024);
    const int fx = p.tile_x, fy = p.tile_y;
    // Tree levels walked from LDS (walk_absorb; 12 bytes per heap slot and tree).  LDS given to the tree tops is taken from the
    // tile, and every tile copies them: 4 KB measured best on the bench workload (10 trees, stride 4: 5 levels 0.175 ms,
    // 6 levels 0.188) and for 20 trees at stride 4 (4 levels 0.157, 5 levels 0.166, 7 levels 0.191).  More trees mean more walks
    // per window, and a smaller stride means more windows per byte of region -- both make a level saved worth more LDS:
    // config 3 (50 trees, stride 2) 6 levels 0.284 ms, 5 levels 0.313, 4 levels 0.343, 2 levels 0.365; config 5 (stride 1) 8 levels.
    // Hence 200 bytes per tree, scaled by (4 / stride)^2, between 4 and 40 KB; then whatever fits beside the tile for nothing.
    g.top_levels = 0;
    if (g.uniform && p.absorb_ok) {
        if (p.top_levels >= 0) g.top_levels = p.top_levels;
        else {
            const double want = 200.0 * p.n_trees * 16.0 / ((double)step * step);
            const size_t top_budget = (size_t)std::min(40.0 * 1024, std::max(4.0 * 1024, want));
            while (g.top_levels < 8 && (size_t)p.n_trees * (2u << g.top_levels) * 12 <= top_budget) ++g.top_levels;
        }
        while (g.top_levels > 0 && (size_t)p.n_trees * (1u << g.top_levels) * 12 > 48 * 1024) --g.top_levels;
    }
    const int top_words = g.uniform && p.absorb_ok ? (int)p.n_trees * (1 << g.top_levels) * 3 : 0;
#ifdef TRAV_THREADS
    long max_win = TRAV_THREADS;
#else
    long max_win = 1024;      // one thread per window position in the gate
#endif
    // single-frame workspace (the live-camera loop of examples/live_prediction.rs:76-86): its tiles cannot fill the chip anyway, so
    // smaller ones -- all walks of a tile in ONE lock-step pass, wider than high (fewer region rows to copy), the frame spread over more CUs -- cost
    // nothing and shorten the frame's critical path (one 320 x 240 frame at stride 1, host sync after each: 91.7 us with tiles of
    // 24 x 17 windows, 90.0 with 20 x 20, 87.9 with 20 x 16, 88.5 with 16 x 12, 89.3 with 12 x 12; 32 x 32, the largest that fits: 95)
    if (p.one_pass && g.uniform && p.absorb_ok) max_win = std::max<long>(16, std::min<long>(max_win, 3200 / std::max<uint32_t>(p.n_trees, 1)));
    long best = -1;
    for (int py = 1; py <= std::min(g.ny, 64); ++py)
        for (int px = 1; px <= std::min(g.nx, 64); ++px) {
            if (px * py > max_win) continue;
            if (fx > 0 && fy > 0 && (px != std::min(fx, g.nx) || py != std::min(fy, g.ny))) continue;
            // uniform path: tiles start on 16-byte boundaries of the box image's planes (direct-to-LDS copy)
            if (rw > 0 && (px & 3) != 0 && px < g.nx && !(fx > 0)) continue;
            // uniform path: the packed rectangle offsets of a compact node are 14-bit
            if (rw > 0 && (long)(sh - rh + 1) * dh_traverse_row_stride(px, step, sw, rw) >= 16384) continue;
            size_t lds = dh_traverse_lds_bytes(px, py, step, sw, sh, top_words, rw, rh);
            if (lds > budget && !(fx > 0 && lds <= 158 * 1024)) continue;
            long score = (long)px * py * 1000 - labs((long)px - py);
            if (p.one_pass && g.uniform && p.absorb_ok) score = (long)px * py * 1000 + (px - py);
            if (score > best) { best = score; g.px = px; g.py = py; g.lds = lds; }
        }
    if (best < 0) {
        // a single position must always fit
        if (rw > 0 && (long)(sh - rh + 1) * dh_traverse_row_stride(1, step, sw, rw) >= 16384) return 1;   // caller retries on the general path
        g.px = g.py = 1;
        g.lds = dh_traverse_lds_bytes(1, 1, step, sw, sh, top_words, rw, rh);
        if (g.lds > 158 * 1024) return rw > 0 ? 1 : dh_fail_(DH_ESIZE, "patch %dx%d with %u trees does not fit LDS", sw, sh, p.n_trees);
    }
    // a tile capped by its 1024 threads (small strides) or by the frame leaves LDS unused: more tree levels fit for nothing
    if (g.uniform && p.absorb_ok && p.top_levels < 0)
        while (g.top_levels < 8 && g.lds + (size_t)p.n_trees * (1u << g.top_levels) * 12 <= budget) {
            g.lds += (size_t)p.n_trees * (1u << g.top_levels) * 12;      // (doubling the table adds its current size)
            ++g.top_levels;
        }
    g.tiles_x = (g.nx + g.px - 1) / g.px;
    g.tiles_y = (g.ny + g.py - 1) / g.py;
    if ((long)g.tiles_x * g.tiles_y > 65535) return dh_fail_(DH_ESIZE, "frame %dx%d needs %ld tiles per frame (limit 65535)", g.w, g.h, (long)g.tiles_x * g.tiles_y);
    g.win_cap = g.tiles_x * g.tiles_y * g.px * g.py;
    g.flag_words = (g.tiles_x * g.tiles_y + 3) / 4;
    g.ss_row = dh_traverse_row_stride(g.px, step, sw, rw);
    if (rw > 0) dh_traverse_swizzle(g.px, step, sw, rw, &g.swz_log2, &g.swz_q, &g.ss_row);
    g.ss_max = g.ss_row * ((g.py - 1) * step + sh + (rw > 0 ? 1 - rh : 1));
    if (rw > 0) {
        g.box_rows = g.h - rh + 1;
        // a row of the image = m planes (same de-interleave as the LDS region) of box_plane words;
        // the slack lets a tile read whole 16-byte groups past its last column
        const int m = 1 << g.swz_log2;
        g.box_plane = ((g.w - rw + 1 + m - 1) / m + 4 + 3) & ~3;
        // one wave yields up to 256 - rw columns (a multiple of 4) of a band of rows; bands are sized so
        // that a batch of a few hundred frames fills the chip once (about 24 waves per frame at VGA)
        const int bw = g.w - rw + 1, ow_max = (kBoxSpan - rw) & ~3;
        g.box_parts = (bw + ow_max - 1) / ow_max;
        g.box_ow = std::min(ow_max, ((bw + g.box_parts - 1) / g.box_parts + 3) & ~3);
        const int band = std::max(1, p.box_band);
        g.box_bands = (g.box_rows + band - 1) / band;
        g.box_oh = ((g.box_rows + g.box_bands - 1) / g.box_bands + 31) & ~31;     // whole 32-row blocks per band (BoxArgs::blk_mask)
        g.box_bands = (g.box_rows + g.box_oh - 1) / g.box_oh;
    }
    return DH_OK;
}

// ------------------------------------------------------------------ small numeric tables
// Mat3<f32>::inv = adjugate / det, element-wise (meancov_estimation.rs:339-352); f32, no FMA
// k_boxsum's bands (dh_host.h).  Measured on MI355X, 640 x 480, 24 x 24 rectangles, 3 parts (tools/experiments/box_bands_sweep.sh,
// profiles/r03_experiments.md): the kernel's time is (rows a wave marches; the rh - 1 rows that only fill the ring count half) x
// (rounds the workgroups need on the chip's slots; a partly filled last round costs half of what it leaves empty).  The rule this
// replaces -- "as many bands as keep every WAVE resident" -- ignored that waves come in workgroups of four: 512 frames took 2 bands
// (6 of 8 wave slots used, 1 024 workgroups on 768 slots: 0.28 ms) where 4 bands take 0.19 ms; 320 frames 0.174 -> 0.134 ms.
int dh_box_bands_(int n, int parts, int rows, int blk, int rh, int wg_slots, int *oh_out) {
    n = std::max(n, 1); parts = std::max(parts, 1); rows = std::max(rows, 1); blk = std::max(blk, 1); wg_slots = std::max(wg_slots, 1);
    const int max_bands = std::max(1, (rows + blk - 1) / blk);
    int best_bands = 1, best_oh = ((rows + blk - 1) / blk) * blk;
    double best = -1.0;
    for (int want = 1; want <= max_bands; ++want) {
        const int oh = (((rows + want - 1) / want + blk - 1) / blk) * blk;        // whole mask blocks per band
        const int bands = (rows + oh - 1) / oh;
        if (bands != want && want > 1) continue;                                   // (the same cut as a smaller `want`)
        const double wgs = (double)n * ((parts * bands + 3) / 4);
        const double r = wgs / wg_slots;
        const double rounds = r <= 1.0 ? 1.0 : r + 0.5 * (std::ceil(r) - r);
        const double cost = (oh + 0.5 * std::max(rh - 1, 0)) * rounds;
        if (best < 0.0 || cost < best - 1e-9) { best = cost; best_bands = bands; best_oh = oh; }   // ties: fewer bands
    }
    if (oh_out) *oh_out = best_oh;
    return best_bands;
}

void dh_mat3_inv_f32_(const float m[9], float o[9]) {
    const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const float det = a * (e * i - f * h) - d * (b * i - c * h) + g * (b * f - c * e);   // :340-342
    o[0] = (e * i - f * h) / det; o[1] = (c * h - b * i) / det; o[2] = (b * f - c * e) / det;   // :348-350
    o[3] = (f * g - d * i) / det; o[4] = (a * i - c * g) / det; o[5] = (c * d - a * f) / det;
    o[6] = (d * h - e * g) / det; o[7] = (b * g - a * h) / det; o[8] = (a * e - b * d) / det;
}

// FullArray3D::build_kernel(20, sigma) (meanshift.rs:228-252), stored in summation order
// (dx*20+dy)*20+dz; the reference indexes kernel[(x+10, y+10, z+10)] = data[z*400 + y*20 + x].
void dh_build_kernel_table_(float sigma, std::vector<float> &k) {
    const int G = DH_MEANSHIFT_KERNEL_SIZE;
    k.resize((size_t)G * G * G);
    for (int x = 0; x < G; ++x)
        for (int y = 0; y < G; ++y)
            for (int z = 0; z < G; ++z) {
                int dx = x - 10, dy = y - 10, dz = z - 10;
                int norm = dx * dx + dy * dy + dz * dz;
                k[((size_t)x * G + y) * G + z] = expf(-1.0f * (float)norm / (2.0f * sigma));
            }
}
// The same weights by squared distance: kernel_function (meanshift.rs:228-232) sees only norm = dx^2 + dy^2 + dz^2, at most
// 3 * 10^2, so 301 values are the whole kernel (k_cluster keeps them in LDS).
void dh_build_kernel_r2_(float sigma, std::vector<float> &k, size_t padded) {
    k.assign(padded, 0.0f);
    for (int norm = 0; norm <= 300 && (size_t)norm < padded; ++norm) k[(size_t)norm] = expf(-1.0f * (float)norm / (2.0f * sigma));
}

// imageproc 0.12.0 filter::gaussian_kernel_f32(sigma) (crate source not in the container; PARITY UNPINNED): radius
// ceil(2 sigma), taps = the zero-mean normal density at 0, 1, ..., radius mirrored, NOT renormalised;
// gaussian(x, r) = ((2.0 * PI).sqrt() * r).recip() * (-x.powi(2) / (2.0 * r.powi(2))).exp(), all in f32.
int dh_blur_taps_(float sigma, std::vector<float> &k) {
    if (!(sigma > 0.0f)) return dh_fail_(DH_EINVAL, "gaussian_blur_f32 needs sigma > 0 (the reference asserts)");
    const float r2 = ceilf(2.0f * sigma);
    if (!(r2 <= 2048.0f)) return dh_fail_(DH_ESIZE, "blur radius %g too large", (double)r2);
    const int radius = (int)r2;
    k.assign((size_t)2 * radius + 1, 0.0f);
    const float norm = 1.0f / (sqrtf(2.0f * 3.14159274101257324f) * sigma);
    for (int i = 0; i <= radius; ++i) {
        const float x = (float)i;
        const float v = norm * expf(-(x * x) / (2.0f * (sigma * sigma)));
        k[radius + i] = v; k[radius - i] = v;
    }
    return DH_OK;
}

// ------------------------------------------------------------------ upload chunking
int dh_chunk_plan_(int m, int stage_chunk, bool single, int cstart[DH_STAGE_EVENTS + 1]) {
    if (m <= 0) { cstart[0] = 0; return 0; }
    if (single) { cstart[0] = 0; cstart[1] = m; return 1; }
    int chunk = std::min(m, std::max(1, stage_chunk));
    chunk = std::max(chunk, (m + DH_STAGE_EVENTS / 2 - 1) / (DH_STAGE_EVENTS / 2));
    int nchunks = 0;
    for (int c0 = 0; c0 < m && nchunks < DH_STAGE_EVENTS;) {
        cstart[nchunks++] = c0;
        const int left = m - c0;
        int c = std::min(chunk, std::max(16, (left + 1) / 2));
        if (nchunks == DH_STAGE_EVENTS || left - c < 8) c = left;
        c0 += c;
    }
    cstart[nchunks] = m;
    return nchunks;
}

// ------------------------------------------------------------------ run-length coded input (BIWI `.bin`, biwi.rs:81-103)
void dh_parallel_for_(int n, int threads, const std::function<void(int)> &fn) {
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::atomic<int> next{0};
    auto body = [&]() { for (int i; (i = next.fetch_add(1)) < n;) fn(i); };
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    try { for (int t = 1; t < threads; ++t) pool.emplace_back(body); } catch (...) { /* fewer workers: the caller's thread finishes the rest */ }
    body();
    for (auto &t : pool) t.join();
}

static inline uint32_t rd32(const uint8_t *q) { uint32_t v; memcpy(&v, q, 4); return v; }   // little-endian host (x86-64)

long dh_rle_scan_(const uint8_t *buf, size_t len, uint32_t W, uint32_t H, DhRun *runs, uint32_t dst0, uint32_t src0, char *err, size_t errn) {
    const size_t total = (size_t)W * H;
    size_t p = 0, pos = 8;
    long cnt = 0;
    if (len < 8) { snprintf(err, errn, "depth payload truncated in the header"); return -1; }
    while (p < total) {
        if (len - pos < 4) { snprintf(err, errn, "depth payload truncated at byte %zu", pos); return -1; }
        const uint32_t n_empty = rd32(buf + pos); pos += 4;
        if ((size_t)n_empty > total - p) { snprintf(err, errn, "run of %u empty pixels overruns the image (reference panics, biwi.rs:92)", n_empty); return -1; }
        p += n_empty;
        if (len - pos < 4) { snprintf(err, errn, "depth payload truncated at byte %zu", pos); return -1; }
        const uint32_t n_full = rd32(buf + pos); pos += 4;
        if ((size_t)n_full > total - p) { snprintf(err, errn, "run of %u pixels overruns the image (reference panics, biwi.rs:97)", n_full); return -1; }
        if ((len - pos) / 2 < n_full) { snprintf(err, errn, "depth payload truncated inside a run at byte %zu", pos); return -1; }
        if (n_full) {
            if (runs) { runs[cnt].dst = dst0 + (uint32_t)p; runs[cnt].src = src0 + (uint32_t)(pos >> 1); }
            ++cnt;
        }
        pos += (size_t)n_full * 2;
        p += n_full;
    }
    return cnt;
}

int dh_rle_plan_(const uint8_t *const *bufs, const size_t *lens, int n, int threads, RlePlan &plan) {
    if (n <= 0) return dh_fail_(DH_EINVAL, "batch size must be positive");
    for (int i = 0; i < n; ++i) {
        if (!bufs[i]) return dh_fail_(DH_EINVAL, "frame %d: NULL payload", i);
        if (lens[i] < 8) return dh_fail_(DH_EINVAL, "frame %d: depth payload truncated in the header", i);      // read_u32 fails (biwi.rs:83-84)
    }
    const uint32_t W = rd32(bufs[0]), H = rd32(bufs[0] + 4);
    if (W == 0 || H == 0 || (uint64_t)W * H > 0x7fffffffull) return dh_fail_(DH_ESIZE, "frame 0: unsupported image size %ux%u", W, H);
    if ((uint64_t)n * W * H > 0xffffffffull) return dh_fail_(DH_ESIZE, "batch of %d frames of %ux%u exceeds 2^32 pixels; split it", n, W, H);
    plan.W = W; plan.H = H;
    plan.blob_off.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) plan.blob_off[i + 1] = plan.blob_off[i] + ((lens[i] + 15) & ~(size_t)15);
    if (plan.blob_off[n] / 2 > 0xffffffffull) return dh_fail_(DH_ESIZE, "payloads exceed 8 GiB; split the batch");
    std::vector<long> counts((size_t)n, 0);
    std::vector<std::string> errs((size_t)n);
    dh_parallel_for_(n, threads, [&](int i) {
        char e[160] = "";
        if (rd32(bufs[i]) != W || rd32(bufs[i] + 4) != H) { snprintf(e, sizeof e, "image is %ux%u, frame 0 is %ux%u", rd32(bufs[i]), rd32(bufs[i] + 4), W, H); counts[i] = -1; }
        else counts[i] = dh_rle_scan_(bufs[i], lens[i], W, H, nullptr, 0, 0, e, sizeof e);
        if (counts[i] < 0) errs[i] = e;
    });
    for (int i = 0; i < n; ++i)
        if (counts[i] < 0) return dh_fail_(DH_EINVAL, "frame %d: %s", i, errs[i].c_str());
    plan.run_begin.assign((size_t)n + 1, 0);
    size_t nruns = 0;
    for (int i = 0; i < n; ++i) {
        nruns += (size_t)counts[i];
        if (nruns > 0xffffffffull) return dh_fail_(DH_ESIZE, "too many runs");
        plan.run_begin[i + 1] = (uint32_t)nruns;
    }
    plan.nruns = nruns;
    return DH_OK;
}

void dh_rle_pack_(const uint8_t *const *bufs, const size_t *lens, int n, int threads, const RlePlan &plan, uint8_t *blob, DhRun *runs) {
    dh_parallel_for_(n, threads, [&](int i) {
        memcpy(blob + plan.blob_off[i], bufs[i], lens[i]);
        char e[8];
        (void)dh_rle_scan_(bufs[i], lens[i], plan.W, plan.H, runs + plan.run_begin[i], (uint32_t)((size_t)i * plan.W * plan.H),
                           (uint32_t)(plan.blob_off[i] >> 1), e, sizeof e);
    });
}
