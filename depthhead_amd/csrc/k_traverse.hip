// k_traverse.hip -- the walk kernel: tile of window positions -> region in LDS -> background gate -> root-to-leaf walks (k_traverse)
//
// One of the kernel translation units of libdepthhead_hip.so (hand-written HIP for gfx950: wave64, 160 KB LDS/CU;
// no MFMA anywhere -- there is no dense contraction on this path).  Overview of the pipeline: dh_api.hip.
#include "dh_device.h"

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
#ifndef TRAV_THREADS
#define TRAV_THREADS 1024
#endif
#define TRAV_WMAX (4096 / TRAV_THREADS)     // lock-step walks per lane of the walk-table path: 4 at 1024 threads (64 VGPRs), 8 at 512 (128)
#define TRAV_WAVES (TRAV_THREADS / WAVE)
#define ROWS_IN_FLIGHT 8

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

// Root-to-leaf walks of the uniform path: work item k = (tree k / n_active, active slot k % n_active);
// a lane takes items tid, tid + 1024, ... W at a time.  Leaf ids go to the tile's segment of the frame's
// window list (wleaf, read by k_emit) and, when the taps are on, to the dense [position][tree] array.  HoughTreeFunctions::binarize
// (houghforest.rs:185-193) on two rectangle sums with the integer test of NodeU, falling back to
// the reference's own f64 arithmetic inside the band the integer test cannot decide.
template <int W>
__device__ __forceinline__ void walk_uniform(const TraverseArgs &a, const uint32_t *sat, char *wleaf, int32_t *dleaf, const uint32_t *active,
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
                store_leaf(wleaf, a.leaf_ls, dst[i], ~cur[i]);
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
__device__ __forceinline__ void walk_absorb(const TraverseArgs &a, const uint32_t *sat, const uint32_t *top, char *wleaf, int32_t *dleaf,
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
                store_leaf(wleaf, a.leaf_ls, dst[i], leaf);
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
__device__ __forceinline__ void walk_general_int(const TraverseArgs &a, const uint32_t *sat, char *wleaf, int32_t *dleaf, const uint32_t *active,
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
                store_leaf(wleaf, a.leaf_ls, dst[i], ~cur[i]);
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
__global__ void __launch_bounds__(TRAV_THREADS, TRAV_THREADS / 128) k_traverse(TraverseArgs a) {
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
    bool nonzero = a.tile_list != nullptr || a.tile_flags[(size_t)frame * (a.tiles_x * a.tiles_y) + tile] == (uint8_t)*a.gen;
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
    char *wleaf = (char *)a.win_leaf + (((size_t)frame * a.win_cap * T + wbase) << a.leaf_ls);   // [tree][win_cap] per frame
    int32_t *dleaf = a.dbg_leaf ? a.dbg_leaf + (size_t)frame * a.nx * a.ny * T : nullptr;
    if (UNI) {
        // W walks per lane, advanced in lock step: their node fetches and box-sum reads are
        // independent, so each lane keeps W dependent-load chains in flight.  W = walks per lane
        // of this tile (at most 4), so one pass covers the tile whenever it has <= 4096 walks.
        // (per WAVE: the last pass over the items is usually filled in part, and a wave without items in it walks one chain
        // fewer -- every lock-step level of a chain is a node gather of the whole wave, whatever its lanes hold)
        const int wave_first = __builtin_amdgcn_readfirstlane(tid & ~(WAVE - 1));
        const int all_passes = (total - wave_first + TRAV_THREADS - 1) / TRAV_THREADS;
        const int per_lane = all_passes > TRAV_WMAX ? (total + TRAV_THREADS - 1) / TRAV_THREADS : all_passes;
        if (GI) {                   // (uniform path: the second template flag selects the absorbing-leaf walk table)
            if (per_lane <= 1) walk_absorb<1>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
            else if (per_lane == 2) walk_absorb<2>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
            else if (per_lane == 3) walk_absorb<3>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
#if TRAV_WMAX >= 8
            else if (per_lane == 4) walk_absorb<4>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
            else if (per_lane == 5) walk_absorb<5>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
            else if (per_lane == 6) walk_absorb<6>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
            else if (per_lane == 7) walk_absorb<7>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
            else walk_absorb<8>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
#else
            else walk_absorb<4>(a, sat, top, wleaf, dleaf, active, agp, n_active, total, cx, ss, T);
#endif
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
            store_leaf(wleaf, a.leaf_ls, t * a.win_cap + slot, ~cur);
            if (dleaf) dleaf[(size_t)agp[slot] * T + t] = ~cur;
        }
    }

    STAMP(2)
}

// Raise the dynamic-LDS limit of the walk kernel: a per-DEVICE attribute, set once per device and process (not allowed during
// stream capture, so dh_predictor_create calls it, with the device current).  Predictors are created concurrently by
// independent host threads (the header allows it): the per-device flags are atomics and a second caller that finds the
// flag clear simply sets the same value again.
#include <algorithm>
#include <atomic>
hipError_t dh_kernels_init(int device) {
    static std::atomic<unsigned long long> done[4];                 // one bit per device id < 256
    const bool tracked = device >= 0 && device < 256;
    if (tracked && (done[device >> 6].load(std::memory_order_acquire) >> (device & 63)) & 1ull) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void *)k_traverse<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_traverse<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_traverse<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_traverse<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = dh_region_init();
    if (e != hipSuccess) return e;
    if (tracked) done[device >> 6].fetch_or(1ull << (device & 63), std::memory_order_release);
    return hipSuccess;
}

hipError_t dh_launch_traverse(const TraverseArgs &a, size_t lds_bytes, hipStream_t s) {
    const int tiles = a.tiles_x * a.tiles_y, fb = (a.n_frames + 7) / 8;
    if (tiles == 0 || fb == 0) return hipSuccess;
    if (tiles > 65535 || fb > 65535) return hipErrorInvalidConfiguration;
    const dim3 grid((unsigned)std::min(8, a.n_frames), tiles, fb);      // (fewer than 8 frames: no workgroups for the frames that are not there)
    if (a.uniform && a.nodes_a) hipLaunchKernelGGL((k_traverse<true, true>), grid, dim3(TRAV_THREADS), lds_bytes, s, a);
    else if (a.uniform) hipLaunchKernelGGL((k_traverse<true, false>), grid, dim3(TRAV_THREADS), lds_bytes, s, a);
    else if (a.nodes_g) hipLaunchKernelGGL((k_traverse<false, true>), grid, dim3(TRAV_THREADS), lds_bytes, s, a);
    else hipLaunchKernelGGL((k_traverse<false, false>), grid, dim3(TRAV_THREADS), lds_bytes, s, a);
    return hipGetLastError();
}
