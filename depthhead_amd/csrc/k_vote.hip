// k_vote.hip -- coarse guess grids (k_vote)
//
// One of the kernel translation units of libdepthhead_hip.so (hand-written HIP for gfx950: wave64, 160 KB LDS/CU;
// no MFMA anywhere -- there is no dense contraction on this path).  Overview of the pipeline: dh_api.hip.
#include "dh_device.h"

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
            // Only the CELL of the 20 x 20 grid is needed here.  When w and h are multiples of 20 (cells are whole pixels wide)
            // the cell of the reference's clamped, truncated quotient x2 (:662-672) is floor(clamp(x2 * 20 / w)), and an
            // approximate quotient decides it whenever it is not next to a cell border.  Pinhole intrinsics: u = x2 * 20 / w =
            // (nx / nz) * (fx * 20 / w) + cx * 20 / w, taken as fma(nx * rcp(nz), kxs, cxs) with the two constants rounded once on
            // the host: v_rcp_f32 (1 ulp), three roundings and the two constants put it within 9.2e-6 of the real u for |u| <= 21
            // (|first term| <= 31), the reference's own four roundings move its quotient by < 7.4e-6 more, and outside [0, 20) both
            // sides clamp into cell 0 / 19 wherever they are.  Quotients within 1e-4 of an integer, and everything not finite
            // (nz = 0, NaN), take the reference's expression and its two IEEE divisions -- 0.04 % of the votes.
            uint32_t idx;
            float ux, uy;
            if (PINNED) {
                const float rc = __builtin_amdgcn_rcpf(nz);
                ux = __builtin_fmaf(__fmul_rn(nx, rc), a.kxs, a.cxs); uy = __builtin_fmaf(__fmul_rn(ny, rc), a.kys, a.cys);
            } else {
                float r[3];
                matvec3(a.k, nx, ny, nz, r);                                      // types.rs:425
                const float rc = __builtin_amdgcn_rcpf(r[2]);
                ux = __fmul_rn(__fmul_rn(r[0], rc), a.sx); uy = __fmul_rn(__fmul_rn(r[1], rc), a.sy);
            }
            const bool near_border = !(fabsf(__fsub_rn(ux, rintf(ux))) > 1.0e-4f) || !(fabsf(__fsub_rn(uy, rintf(uy))) > 1.0e-4f);
            if (a.cell_fast && !near_border) {
                const float cxf = fminf(fmaxf(ux, 0.0f), 19.5f), cyf = fminf(fmaxf(uy, 0.0f), 19.5f);
                idx = (uint32_t)cyf * DH_GRID + (uint32_t)cxf;
            } else {
                float r[3];
                if (PINNED) {
                    r[0] = __fadd_rn(__fmul_rn(nx, a.k[0]), __fmul_rn(nz, a.k[2]));
                    r[1] = __fadd_rn(__fmul_rn(ny, a.k[4]), __fmul_rn(nz, a.k[5]));
                    r[2] = __fadd_rn(nz, 0.0f);                                   // (x * 0 + y * 0) + z * 1: -0 becomes +0
                } else {
                    matvec3(a.k, nx, ny, nz, r);                                  // types.rs:425
                }
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
// LH: the batch has a leaf histogram (VoteArgs::leaf_hits): rotation cells come from the leaves that voted, not from the hit
// records (an instance of its own: the other path's registers would cost the 80-VGPR instance a wave per SIMD).
template <bool TAB, bool PIN, bool LH>
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
    if (n == 0 || (h0 >= h1 && !LH)) return;       // with the leaf histogram every slice also owns a share of the leaves
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
        const uint4 rr = LH ? make_uint4(0u, 0u, 0u, 0u) : *(const uint4 *)(hr + i);   // rotation cells: only without the leaf histogram
        const uint32_t v = (uint32_t)b1.z, fc = (uint32_t)b1.w;
        if (!LH && (fc & LF_ROT))
            for (uint32_t r = rr.z + sub; r < rr.z + (rr.w >> 16); r += VOTE_SUB) { const uint32_t c = a.f.rough_cell[r]; atomicAdd(&rot[c & 0xffffu], v * (c >> 16)); }   // :636
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
    if (LH) {
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
            for (uint32_t r = t2.w; r < t2.w + (t3.x >> 16); ++r) { const uint32_t c = a.f.rough_cell[r]; atomicAdd(&rot[c & 0xffffu], cv * (c >> 16)); }   // :636
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
    // 8 400 cells with atomics, so 128 slices of one frame still cost less than a mostly idle chip: one 320 x 240 frame at stride 1
    // 18.6 / 12.7 / 10.2 / 8.0 / 8.0 us with at most 16 / 32 / 64 / 128 / 256 slices)
    const uint32_t slices = a.n_frames >= 128 ? VOTE_SLICES : std::min(128u, std::max((uint32_t)VOTE_SLICES, 1024u / (uint32_t)a.n_frames));
    const dim3 grid(slices, a.n_frames), block(VOTE_THREADS);
    VoteArgs b = a;
    b.cell_fast = a.cell_fast && a.w % DH_GRID == 0 && a.h % DH_GRID == 0 && a.w > 0 && a.h > 0;
    b.sx = (float)DH_GRID / (float)a.w; b.sy = (float)DH_GRID / (float)a.h;
    b.kxs = a.k[0] * b.sx; b.cxs = a.k[2] * b.sx; b.kys = a.k[4] * b.sy; b.cys = a.k[5] * b.sy;   // (one f32 rounding each: see vote_positions)
#define VOTE_LAUNCH(TAB_, PIN_)                                                                       \
    do {                                                                                             \
        if (b.leaf_hits) hipLaunchKernelGGL((k_vote<TAB_, PIN_, true>), grid, block, 0, s, b);       \
        else hipLaunchKernelGGL((k_vote<TAB_, PIN_, false>), grid, block, 0, s, b);                  \
    } while (0)
    if (a.w <= VOTE_TAB && a.h <= VOTE_TAB) {
        if (pin) VOTE_LAUNCH(true, true);
        else VOTE_LAUNCH(true, false);
    } else {
        if (pin) VOTE_LAUNCH(false, true);
        else VOTE_LAUNCH(false, false);
    }
#undef VOTE_LAUNCH
    return hipGetLastError();
}
