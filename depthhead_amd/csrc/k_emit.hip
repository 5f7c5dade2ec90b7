// k_emit.hip -- probability gate and hit records (k_emit)
//
// One of the kernel translation units of libdepthhead_hip.so (hand-written HIP for gfx950: wave64, 160 KB LDS/CU;
// no MFMA anywhere -- there is no dense contraction on this path).  Overview of the pipeline: dh_api.hip.
#include "dh_device.h"

// ================================================================== k_emit
// One thread per active window of the list k_traverse wrote: mean leaf probability in tree order
// (prediction.rs:582-584), the > 0.7 gate, and one self-contained hit record per voting leaf
// (consumed by k_vote and k_cluster).  No LDS, no barriers: the three dependent global round trips
// (leaf probabilities, the frame's hit counter, the leaf templates) that used to end every
// k_traverse workgroup are hidden here by plain occupancy.
#define EMIT_THREADS 256
// EB = trees handled per batch of gathers: the smallest instance that holds all T trees keeps the registers (and with them the
// occupancy of this latency-bound kernel) in proportion to the forest: 62 VGPRs at EB = 8, 80 at 10, 122 at 16.
template <int EB, int LS>      // LS: log2 of the window list's leaf entry size (dh_device.h: load_leaf)
__global__ void __launch_bounds__(EMIT_THREADS) k_emit(EmitArgs a) {
    const int frame = blockIdx.y, lane = threadIdx.x & (WAVE - 1);
    // every reader of this batch's tile flags (k_tile_list, k_traverse) has run: the next batch of this kernel sequence takes
    // the next tag, 1 .. 255 (BoxArgs::gen)
    if (a.gen && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.gen = *a.gen % 255u + 1u;
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
    constexpr int ls = LS;
    const char *wl = (const char *)a.win_leaf + (((size_t)frame * a.win_cap * T + w) << ls);
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
            for (int k = 0; k < EB; ++k) l[k] = load_leaf(wl, ls, (size_t)min(t0 + k, T - 1) * a.win_cap);
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
    const char *wlf = (const char *)a.win_leaf + (((size_t)frame * a.win_cap * T) << ls);
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
                    const uint32_t lf = a.f.leaf_flags[load_leaf(wlf, ls, (size_t)t * a.win_cap + sw)];
                    if ((lf & LF_PROB) && (lf & (LF_ROT | LF_OFF))) { if (r == h) { src = sl; tree = t; } ++r; }
                }
            }
        }
        const uint32_t h = chunk + (uint32_t)lane;
        const int sw = __shfl(wi, src);
        const float p0 = __shfl(q0, src), p1 = __shfl(q1, src), p2 = __shfl(q2, src);
        const uint32_t o = base + h;
        if (h < wave_total && o < a.hits_cap) {
            const uint32_t lid = load_leaf(wlf, ls, (size_t)tree * a.win_cap + sw);
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
#define EMIT_LAUNCH(EB_)                                                                       \
    do {                                                                                       \
        if (a.leaf_ls == 1) hipLaunchKernelGGL((k_emit<EB_, 1>), grid, block, 0, s, a);        \
        else hipLaunchKernelGGL((k_emit<EB_, 2>), grid, block, 0, s, a);                       \
    } while (0)
    if (T <= 4) EMIT_LAUNCH(4);
    else if (T <= 8) EMIT_LAUNCH(8);
    else if (T <= 10) EMIT_LAUNCH(10);
    else if (T <= 12) EMIT_LAUNCH(12);
    else EMIT_LAUNCH(16);
#undef EMIT_LAUNCH
    return hipGetLastError();
}
