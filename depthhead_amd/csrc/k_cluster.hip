// k_cluster.hip -- initial guesses and both mean shifts -> pose (k_cluster, k_region), vote-dump tap
//
// One of the kernel translation units of libdepthhead_hip.so (hand-written HIP for gfx950: wave64, 160 KB LDS/CU;
// no MFMA anywhere -- there is no dense contraction on this path).  Overview of the pipeline: dh_api.hip.
#include "dh_device.h"

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
// everything off that chain is parallel: per-row occupancy masks of the region (built once per region) name the
// non-zero cells, a prefix sum over the window's 400 rows places them, the rows' threads write the products.
// What the kernel costs is the rotation workgroup's chain of DEPENDENT round trips and barriers (guess -> list of the
// leaves that voted -> their cells -> three or four sums), not work: see DESIGN.md section 4 and profiles/r03_cluster_phases.txt.
#define CL_THREADS 1024
#define CL_WAVES (CL_THREADS / WAVE)
#define CL_PROD_CAP 384         // products staged per pass (x4 floats = 6 KB)
#define CL_LIST (CL_PROD_CAP * 4) // survivors of the region gathers' bounding-box tests listed in `prod` (1536)
#define RG 26                   // region edge; the window may sit at offsets 0..RG-20 inside it
#define RG3 (RG * RG * RG)
static_assert(RG3 == DH_REGION_CELLS, "dh_internal.h: DH_REGION_CELLS");
// k_region's block of an accumulator in global memory (small batches with many hit records per frame): SRG^3 cells
// around the initial guess, from which k_cluster cuts its regions -- the window may travel SPAD cells either way
// on every axis before a region has to be gathered from the hit records again
#define SRG 64
#define SRG3 (SRG * SRG * SRG)
#define SPAD ((SRG - RG) / 2)
static_assert(SRG3 == DH_SUPER_CELLS, "dh_internal.h: DH_SUPER_CELLS");
// ... and of the rotation accumulator (first RRG^3 words of its slot): gathered in LDS, one workgroup per CU
#define RRG 32
#define RRG3 (RRG * RRG * RRG)
#define RPAD ((RRG - RG) / 2)

// exists c in [lo,hi] and d in [0,len) with c == start + d (i32 wrapping, like the reference's
// release-mode `pos + offset`)?
__device__ __forceinline__ bool range_hits_span(int32_t lo, int32_t hi, int32_t start, uint32_t len) {
    uint32_t u = (uint32_t)start - (uint32_t)lo;
    return u <= (uint32_t)hi - (uint32_t)lo || u >= (uint32_t)(1u - len);
}

// Position votes of hit record i that fall into the region: lane `sub` of `nsub` takes the leaf's offset votes
// sub, sub + nsub, ... (prediction.rs:647-667).
template <int EDGE>
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
        if (dx < EDGE && dy < EDGE && dz < EDGE) atomicAdd(&region[(dx * EDGE + dy) * EDGE + dz], v);
    }
}

// Shared-memory carve-up of k_cluster / k_region.
struct ClShared {
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
        uint32_t gv[(DH_GRID3 + CL_THREADS - 1) / CL_THREADS];          // a thread's eight cells of the 20^3 grid: all loads in flight, one round trip
#pragma unroll
        for (int j = 0; j < (DH_GRID3 + CL_THREADS - 1) / CL_THREADS; ++j) { const int i = tid + j * CL_THREADS; gv[j] = i < ncell ? g[i] : 0u; }
#pragma unroll
        for (int j = 0; j < (DH_GRID3 + CL_THREADS - 1) / CL_THREADS; ++j) {
            const unsigned long long k = ((unsigned long long)gv[j] << 32) | (uint32_t)(~(uint32_t)(tid + j * CL_THREADS));
            if (gv[j] && k > best) best = k;
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

// Adds to the (zeroed) block `region` of EDGE^3 cells with origin `org` -- k_cluster's LDS region, or k_region's block of the
// accumulator in global memory -- every vote of accumulator `which` that falls into it, from the hit
// records [h0, h1) of the frame -- or, for rotation votes of forests with a leaf histogram, from the leaves [l0, l1).
// Integer atomics: exact and order-free, so any split of the ranges over workgroups sums to the same block.
template <int EDGE>
__device__ __forceinline__ void cl_gather(const ClusterArgs &a, const int which, const int frame, const int32_t org[3],
                                          const uint32_t h0, const uint32_t h1, const uint32_t l0, const uint32_t l1, const ClShared sh,
                                          uint32_t *region) {
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
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
                const bool keep = i < h1 && (fc & LF_OFF) && range_hits_span(b0[j].x, b0[j].w, org[0], EDGE) &&
                                  range_hits_span(b0[j].y, b1[j].x, org[1], EDGE) && range_hits_span(b0[j].z, b1[j].y, org[2], EDGE);
                const unsigned long long bal = __ballot(keep);             // one LDS atomic per wave, ranks from the ballot
                uint32_t wb = 0;
                if (lane == 0 && bal) wb = atomicAdd(&s_total, (uint32_t)__popcll(bal));
                wb = __shfl(wb, 0);
                if (keep) {
                    const uint32_t slot = wb + (uint32_t)__popcll(bal & lanemask_lt());
                    if (slot < CL_LIST) list[slot] = i;
                    else cluster_add_votes<EDGE>(a, region, hits, box, i, 0u, 1u, org);   // list full: this thread takes the record alone
                }
            }
        }
        __syncthreads();
        const uint32_t np = min(s_total, (uint32_t)CL_LIST);
        for (uint32_t k = tid; k < np * 16u; k += CL_THREADS) cluster_add_votes<EDGE>(a, region, hits, box, list[k >> 4], k & 15u, 16u, org);
        __syncthreads();
    } else if (a.leaf_hits) {
        // Rotation votes depend only on the leaf (prediction.rs:601-636): the accumulator is
        // sum over leaves of (times the leaf voted) x (its distinct cells), so the gather walks
        // the leaves that voted at all instead of every hit -- u32 wrap-around makes
        // hits * v * mult the same residue as that many separate adds.
        const uint32_t *lh = a.leaf_hits + (size_t)frame * a.f.n_leaves;
        uint32_t *list = (uint32_t *)prod;                    // same two-step scheme as the position gather
        // Two steps, like the position gather: (1) the leaves that voted and whose cells can reach the block are listed -- a thread's
        // four histogram loads (leaves tid, tid + 1024, ...) are in flight together, then the four directory entries {bounding
        // box, first cell, cells} of those that voted: two round trips per 4096 leaves; (2) eight lanes share each listed leaf and
        // take its cells eight apart, and a thread has four such (leaf, eighth) items in flight: directory entry, valtoadd and
        // histogram count of all four in one round trip, their cells in the next.  What a lane pays for here is the number of
        // DEPENDENT round trips (bench workload: ~1 000 listed leaves of ~10 cells; one item after the other, sixteen lanes per
        // leaf and the leaf's template fetched per item were ~32 round trips per thread: 0.023 of k_cluster's 0.059 ms).
        // A leaf that finds the list full is handled by its thread at once.
        auto leaf_alone = [&](const uint32_t l, const uint4 d) {
            const uint32_t v = lh[l] * a.f.leaf_v[l];                        // times the leaf voted x valtoadd
            for (uint32_t q = d.z; q < d.z + d.w; ++q) {
                const uint32_t b = a.f.rot_bin[q], vm = v * (b >> 24);                                            // prediction.rs:635
                uint32_t dx = (b & 255u) - (uint32_t)org[0];
                uint32_t dy = ((b >> 8) & 255u) - (uint32_t)org[1];
                uint32_t dz = ((b >> 16) & 255u) - (uint32_t)org[2];
                if (dx < EDGE && dy < EDGE && dz < EDGE) atomicAdd(&region[(dx * EDGE + dy) * EDGE + dz], vm);
            }
        };
        constexpr uint32_t LR = 4, LL = 8, LI = 4;                             // leaves per thread and scan round; lanes per listed leaf; items in flight
        for (uint32_t c0 = l0; c0 < l1; c0 += LR * CL_THREADS) {
            if (tid == 0) s_total = 0;
            __syncthreads();
            uint32_t cnt[LR];
            uint4 d[LR];
#pragma unroll
            for (uint32_t r = 0; r < LR; ++r) { const uint32_t l = c0 + r * CL_THREADS + tid; cnt[r] = l < l1 ? lh[l] : 0u; }
#pragma unroll
            for (uint32_t r = 0; r < LR; ++r) {
                d[r] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
                if (cnt[r]) d[r] = a.f.rot_dir[c0 + r * CL_THREADS + tid];
            }
#pragma unroll
            for (uint32_t r = 0; r < LR; ++r) {                                // (uniform trip count: the ballot needs every lane)
                const uint32_t l = c0 + r * CL_THREADS + tid, bl = d[r].x, bh = d[r].y;
                const bool keep = bl != 0xFFFFFFFFu && range_hits_span((int32_t)(bl & 255u), (int32_t)(bh & 255u), org[0], EDGE) &&
                                  range_hits_span((int32_t)((bl >> 8) & 255u), (int32_t)((bh >> 8) & 255u), org[1], EDGE) &&
                                  range_hits_span((int32_t)((bl >> 16) & 255u), (int32_t)((bh >> 16) & 255u), org[2], EDGE);
                const unsigned long long bal = __ballot(keep);             // one LDS atomic per wave, ranks from the ballot
                uint32_t wb = 0;
                if (lane == 0 && bal) wb = atomicAdd(&s_total, (uint32_t)__popcll(bal));
                wb = __shfl(wb, 0);
                if (keep) {
                    const uint32_t slot = wb + (uint32_t)__popcll(bal & lanemask_lt());
                    if (slot < CL_LIST) list[slot] = l;
                    else leaf_alone(l, d[r]);
                }
            }
            __syncthreads();
            if (KNOB_STOP((a.stop & 15) == 5)) return;
            const uint32_t items = min(s_total, (uint32_t)CL_LIST) * LL;
            for (uint32_t k0 = tid; k0 < items; k0 += LI * CL_THREADS) {
                uint32_t q[LI], q1[LI], vm[LI];
                {
                    uint4 e[LI];
                    uint32_t hc[LI], lv[LI];
#pragma unroll
                    for (uint32_t u = 0; u < LI; ++u) {
                        const uint32_t k = k0 + u * CL_THREADS, l = list[(k < items ? k : k0) / LL];
                        e[u] = a.f.rot_dir[l]; hc[u] = lh[l]; lv[u] = a.f.leaf_v[l];
                    }
#pragma unroll
                    for (uint32_t u = 0; u < LI; ++u) {
                        const uint32_t k = k0 + u * CL_THREADS;
                        q[u] = e[u].z + (k & (LL - 1u)); q1[u] = k < items ? e[u].z + e[u].w : 0u; vm[u] = hc[u] * lv[u];
                    }
                }
                for (;;) {
                    uint32_t b[LI];
#pragma unroll
                    for (uint32_t u = 0; u < LI; ++u) b[u] = a.f.rot_bin[q[u] < q1[u] ? q[u] : 0u];
                    bool more = false;
#pragma unroll
                    for (uint32_t u = 0; u < LI; ++u) {
                        uint32_t dx = (b[u] & 255u) - (uint32_t)org[0];
                        uint32_t dy = ((b[u] >> 8) & 255u) - (uint32_t)org[1];
                        uint32_t dz = ((b[u] >> 16) & 255u) - (uint32_t)org[2];
                        if (q[u] < q1[u] && dx < EDGE && dy < EDGE && dz < EDGE) atomicAdd(&region[(dx * EDGE + dy) * EDGE + dz], vm[u] * (b[u] >> 24));   // prediction.rs:635
                        q[u] += LL;
                        more = more || q[u] < q1[u];
                    }
                    if (!more) break;
                }
            }
            __syncthreads();
        }
    } else {
        // (forests without a leaf histogram: every thread takes its own records' rotation cells.  The two-step survivor list
        // of the position gather was tried here in round 3 and lost: 0.130 vs 0.119 ms on the 35 k-leaf forest -- a record has
        // at most a few dozen distinct cells, and the loads of a thread's loop are independent.)
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
                if (!range_hits_span((int32_t)(bl & 255u), (int32_t)(bh & 255u), org[0], EDGE)) continue;
                if (!range_hits_span((int32_t)((bl >> 8) & 255u), (int32_t)((bh >> 8) & 255u), org[1], EDGE)) continue;
                if (!range_hits_span((int32_t)((bl >> 16) & 255u), (int32_t)((bh >> 16) & 255u), org[2], EDGE)) continue;
                const uint32_t v = vv[j], q1 = r[j].z + (r[j].w & 0xffffu);
                for (uint32_t q0 = r[j].z; q0 < q1; q0 += 4u) {             // four of the record's cells in flight (one load each)
                    uint32_t b[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) b[u] = a.f.rot_bin[min(q0 + u, q1 - 1u)];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        uint32_t dx = (b[u] & 255u) - (uint32_t)org[0];
                        uint32_t dy = ((b[u] >> 8) & 255u) - (uint32_t)org[1];
                        uint32_t dz = ((b[u] >> 16) & 255u) - (uint32_t)org[2];
                        if (q0 + u < q1 && dx < EDGE && dy < EDGE && dz < EDGE) atomicAdd(&region[(dx * EDGE + dy) * EDGE + dz], v * (b[u] >> 24));   // prediction.rs:635
                    }
                }
            }
        }
    }
}

#ifdef DH_PROFILING_KNOBS
#define CSTAMP(k)                                                                               \
    if (a.dbg_stamps && which == 1 && tid == 0) {                                               \
        unsigned long long t_ = clock64();                                                      \
        s_st[k] += t_ - t_prev;                                                                 \
        t_prev = t_;                                                                            \
    }
#else
#define CSTAMP(k)
#endif

// SUP: the batch has blocks of both accumulators in global memory (k_region; a.pre_region != NULL); a separate instance so that
// the code of batches without them keeps its registers.
template <bool SUP>
__global__ void __launch_bounds__(CL_THREADS, 8) k_cluster(ClusterArgs a) {
    __shared__ uint32_t region[RG3];
    __shared__ __attribute__((aligned(16))) float prod[CL_PROD_CAP * 4];
    __shared__ uint32_t cnt[CL_WAVES];
    __shared__ uint32_t rowmask[RG * RG];   // per (x, y) row of the region: bit z set = cell (x, y, z) is non-zero
    __shared__ unsigned long long red64[CL_WAVES];
    __shared__ uint32_t red32[CL_WAVES];
    __shared__ int32_t s_pos[3];
    __shared__ float s_acc[4];
    __shared__ uint32_t s_total;
    __shared__ float s_kr2[DH_KERN_R2];     // Gaussian weights by squared distance: no global load inside a weighted sum

    const int which = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;

    const ClShared sh{prod, red64, red32, s_pos, &s_total};
    if (tid < DH_KERN_R2) s_kr2[tid] = a.kern_r2[tid];        // (visible after the barriers of the initial guess)
    for (int i = tid; i < RG3; i += CL_THREADS) region[i] = 0;   // (likewise: the first gather finds its region zeroed)
    if (KNOB_STOP((a.stop >> 4) == which + 1)) return;          // (profiling twin: DH_CL_STOP = 16 / 32 skips one accumulator)
#ifdef DH_PROFILING_KNOBS
    __shared__ unsigned long long s_st[16];                     // cycles per phase, thread 0 of the rotation workgroups; flushed at the end
    if (tid < 16) s_st[tid] = tid == 15 ? 1ull : 0ull;
    __syncthreads();
    unsigned long long t_prev = a.dbg_stamps ? clock64() : 0ull;
#endif
    cl_initial_guess(a, which, frame, sh);
    __syncthreads();
    CSTAMP(0)
    int32_t pos[3] = {s_pos[0], s_pos[1], s_pos[2]};
    if (a.dbg_guess && tid < 3) a.dbg_guess[(size_t)frame * 6 + which * 3 + tid] = pos[tid];
    int32_t *trace = a.dbg_trace ? a.dbg_trace + ((size_t)which * a.n_frames + frame) * (a.iterations + 1) * 3 : nullptr;
    if (trace && tid < 3) trace[tid] = pos[tid];

    if (KNOB_STOP((a.stop & 15) == 1)) return;
    // ---------------- mean shift (meanshift.rs:328-407)
    uint32_t n_hits = a.hit_count[frame];
    if (n_hits > a.hits_cap) n_hits = a.hits_cap;
    int32_t org[3] = {0, 0, 0};     // region origin (cell coordinates of region[0])
    bool have_region = false, clean = true;   // clean: the region holds zeros (from the start; not after a gather)
    uint32_t steps = 0;
    for (uint32_t it = 0; it < a.iterations; ++it) {
        // window offset inside the region; the region is valid while 0 <= woff <= RG-20 on every axis
        uint32_t wo0 = (uint32_t)pos[0] - 10u - (uint32_t)org[0], wo1 = (uint32_t)pos[1] - 10u - (uint32_t)org[1],
                 wo2 = (uint32_t)pos[2] - 10u - (uint32_t)org[2];
        if (!have_region || wo0 > RG - 20 || wo1 > RG - 20 || wo2 > RG - 20) {
            // ---- (re)build the region centred on the window
            have_region = true;
            __syncthreads();                       // previous iteration's readers are done
            // k_region (several workgroups per frame) built the cells of both accumulators around the initial guess (s_pos still
            // holds it): a region that holds the window and lies inside that block -- centred on the window where the block allows,
            // pushed back inside it otherwise -- is cut out of the block; only a window that has left the block is gathered from
            // the records (any region that holds the window gives the same sums: the region is a cache of the accumulator's cells)
            const int32_t edge = which == 0 ? SRG : RRG, pad = which == 0 ? SPAD : RPAD;
            int32_t dc[3];
            uint32_t wo[3];
            bool fits = SUP && n_hits >= a.pre_min_hits;
            for (int k = 0; k < 3; ++k) {
                // offset of the centred region's origin from the block's (garbage if the window is nowhere near: then nothing fits)
                const int32_t d = (int32_t)((uint32_t)pos[k] - (uint32_t)s_pos[k] + (uint32_t)pad);
                dc[k] = min(max(d, 0), edge - RG);
                wo[k] = (uint32_t)d + (uint32_t)((RG - 20) / 2) - (uint32_t)dc[k];
                fits = fits && wo[k] <= RG - 20;
            }
            if (fits) {
                for (int k = 0; k < 3; ++k) org[k] = (int32_t)((uint32_t)s_pos[k] - 10u - (uint32_t)((RG - 20) / 2) - (uint32_t)pad + (uint32_t)dc[k]);
                wo0 = wo[0]; wo1 = wo[1]; wo2 = wo[2];
                clean = false;
                const uint32_t *sup = a.pre_region + ((size_t)frame * 2 + which) * SRG3 + ((size_t)dc[0] * edge + dc[1]) * edge + dc[2];
                for (int i = tid; i < RG3; i += CL_THREADS) {
                    const uint32_t dz = (uint32_t)i % RG, dy = ((uint32_t)i / RG) % RG, dx = (uint32_t)i / (RG * RG);
                    region[i] = sup[(dx * edge + dy) * edge + dz];
                }
            } else {
                for (int k = 0; k < 3; ++k) org[k] = (int32_t)((uint32_t)pos[k] - 10u - (uint32_t)((RG - 20) / 2));
                wo0 = wo1 = wo2 = (RG - 20) / 2;
                if (!clean) {
                    for (int i = tid; i < RG3; i += CL_THREADS) region[i] = 0;
                    __syncthreads();
                }
                clean = false;
                CSTAMP(1)
                if (KNOB_STOP((a.stop & 15) == 4)) return;
                cl_gather<RG>(a, which, frame, org, 0u, n_hits, 0u, a.f.n_leaves, sh, region);
            }
            // the region's occupancy masks (one thread per (x, y) row)
            __syncthreads();
            if (tid < RG * RG) {
                uint32_t m = 0;
#pragma unroll 2
                for (int z = 0; z < RG; ++z) m |= (region[tid * RG + z] != 0u ? 1u : 0u) << z;
                rowmask[tid] = m;
            }
            __syncthreads();
        }
        CSTAMP(2)
        if (KNOB_STOP((a.stop & 15) == 2)) return;
        // ---- the window's non-zero cells in the reference's summation order (x, then y, then z: meanshift.rs:344-346), through the
        // region's occupancy masks: thread t < 400 owns window row (x, y) = (t / 20, t % 20), whose non-zero cells are the set bits of
        // one mask word; a prefix sum of the rows' counts gives every cell its place, and the row's thread writes the products
        // of its cells (the Gaussian weight comes from LDS).  A sparse window (bench workload: 48 non-zero cells of 8 000) costs three
        // barriers and a few instructions per row instead of two sweeps over all 8 000 cells (1.9 + 2.2 of the 5.1 us of a weighted sum).
        uint32_t mask = 0, rowbase = 0;
        if (tid < 400) {
            const uint32_t dx = (uint32_t)tid / 20u, dy = (uint32_t)tid - dx * 20u;
            rowbase = (wo0 + dx) * RG + wo1 + dy;
            mask = (rowmask[rowbase] >> wo2) & 0xFFFFFu;
        }
        const uint32_t nrow = (uint32_t)__popc(mask);
        const uint32_t incl = wave_incl_scan(nrow);
        if (lane == WAVE - 1) cnt[wave] = incl;          // (waves 7 .. 15 hold no rows: zero)
        __syncthreads();
        CSTAMP(3)
        uint32_t pre = incl - nrow, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < 7; ++w2) { const uint32_t cw = cnt[w2]; total += cw; if (w2 < wave) pre += cw; }
        float acc = 0.0f;                  // (threads 0..3) num.x, num.y, num.z, den
        for (uint32_t base = 0; base < total; base += CL_PROD_CAP) {
            if (pre < base + CL_PROD_CAP && pre + nrow > base) {
                const uint32_t dx = (uint32_t)tid / 20u, dy = (uint32_t)tid - dx * 20u;
                const float ax = (float)(int32_t)((uint32_t)pos[0] + dx - 10u);                     // meanshift.rs:373-375
                const float ay = (float)(int32_t)((uint32_t)pos[1] + dy - 10u);
                const int32_t ex = (int32_t)dx - 10, ey = (int32_t)dy - 10;
                uint32_t k = pre;
                for (uint32_t mm = mask; mm; mm &= mm - 1u, ++k) {
                    if (k < base) continue;
                    if (k >= base + CL_PROD_CAP) break;
                    const uint32_t dz = (uint32_t)__ffs((int)mm) - 1u;
                    const uint32_t fv = region[rowbase * RG + wo2 + dz];
                    const int32_t ez = (int32_t)dz - 10;
                    const float w = __fmul_rn(s_kr2[ex * ex + ey * ey + ez * ez], (float)fv);       // :228-232, :370-379
                    const float az = (float)(int32_t)((uint32_t)pos[2] + dz - 10u);
                    *(float4 *)(prod + (k - base) * 4) = make_float4(__fmul_rn(ax, w), __fmul_rn(ay, w), __fmul_rn(az, w), w);
                }
            }
            __syncthreads();
            CSTAMP(5)
            if (tid < 4) {   // the sequential chain: acc = acc + prod[i], in cell order
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
                if (base + CL_PROD_CAP >= total) s_acc[tid] = acc;
            }
            __syncthreads();
            CSTAMP(7)
        }
        if (total == 0) {                  // an empty window: den == 0 (nothing was summed above)
            if (tid < 4) s_acc[tid] = 0.0f;
            __syncthreads();
        }
        if (KNOB_STOP((a.stop & 15) == 3)) return;
        const float den = s_acc[3];
        if (den == 0.0f) break;                                                              // :385-388
        int32_t np0 = f32_as_i32(__fdiv_rn(s_acc[0], den)), np1 = f32_as_i32(__fdiv_rn(s_acc[1], den)),
                np2 = f32_as_i32(__fdiv_rn(s_acc[2], den));                                   // :391-394
        const bool fixed = np0 == pos[0] && np1 == pos[1] && np2 == pos[2];
        pos[0] = np0; pos[1] = np1; pos[2] = np2;
        steps++;
#ifdef DH_PROFILING_KNOBS
        if (a.dbg_stamps && which == 1 && tid == 0) s_st[14] += 1ull;
#endif
        CSTAMP(8)
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
#ifdef DH_PROFILING_KNOBS
    if (a.dbg_stamps && which == 1 && tid == 0)
        for (int k = 0; k < 16; ++k) atomicAdd(&a.dbg_stamps[k], s_st[k]);
#endif
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
    if (a.pre_region) hipLaunchKernelGGL(k_cluster<true>, dim3(2, a.n_frames), dim3(CL_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k_cluster<false>, dim3(2, a.n_frames), dim3(CL_THREADS), 0, s, a);
    return hipGetLastError();
}

// ================================================================== k_region
// Small batches with many hit records per frame (large forests, stride 1-2: 60-90 k records per frame at BASELINE config 3):
// one workgroup streaming a whole frame's records every time the mean-shift window leaves its 26^3 region is the slowest thing
// in the step while most CUs idle.  Here several workgroups per (frame, accumulator) build, once, in global memory (zeroed per
// batch by the host) the SRG^3 cells of the position accumulator and the RRG^3 cells of the rotation accumulator around the
// initial guesses: workgroup (slice, frame) of accumulator WHICH recomputes the initial guess (a few microseconds) and adds the
// votes of its share of the records (or leaves) that fall into the block, exactly as k_cluster's own gather would (integer
// atomics: the sum over the slices is the reference's accumulator on those cells, prediction.rs:635, :667).  k_cluster then cuts
// the regions it needs out of the blocks: on the config-3 workload the position window drifts a cell per iteration on some
// frames (up to 18 cells in 20 iterations) and the rotation window up to 4 cells, and every 26^3 region a window outgrew used to
// cost a scan of all the frame's records by one workgroup (0.55 ms of that step for one rotation rebuild).
template <int WHICH>
__global__ void __launch_bounds__(CL_THREADS, 8) k_region(ClusterArgs a) {
    extern __shared__ uint32_t block[];                             // WHICH == 1: [RRG3]
    __shared__ __attribute__((aligned(16))) float prod[CL_PROD_CAP * 4];
    __shared__ unsigned long long red64[CL_WAVES];
    __shared__ uint32_t red32[CL_WAVES];
    __shared__ int32_t s_pos[3];
    __shared__ uint32_t s_total;
    const int slice = blockIdx.x, frame = blockIdx.y, which = WHICH, tid = threadIdx.x;
    const ClShared sh{prod, red64, red32, s_pos, &s_total};
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
    uint32_t *pre = a.pre_region + ((size_t)frame * 2 + which) * SRG3;
    int32_t org[3];
    if (WHICH == 0) {
        // position votes: few of a frame's records reach the block and its cells are mostly empty: straight into global memory
        for (int k = 0; k < 3; ++k) org[k] = (int32_t)((uint32_t)s_pos[k] - 10u - (uint32_t)((RG - 20) / 2) - (uint32_t)SPAD);
        cl_gather<SRG>(a, which, frame, org, h0, h1, l0, l1, sh, pre);
    } else {
        // rotation votes are dense around the guess (every record of a frame reaches the block; as scattered global atomics
        // they took 1.16 ms on BASELINE config 3 against 0.12 ms this way): gathered in LDS, then one coalesced flush of the
        // non-zero cells
        for (int k = 0; k < 3; ++k) org[k] = (int32_t)((uint32_t)s_pos[k] - 10u - (uint32_t)((RG - 20) / 2) - (uint32_t)RPAD);
        for (int i = tid; i < RRG3; i += CL_THREADS) block[i] = 0;
        __syncthreads();
        cl_gather<RRG>(a, which, frame, org, h0, h1, l0, l1, sh, block);
        __syncthreads();
        for (int i = tid; i < RRG3; i += CL_THREADS) {
            const uint32_t v = block[i];
            if (v) atomicAdd(&pre[i], v);
        }
    }
}

// (the rotation instance takes 128 KB of dynamic LDS: a per-device attribute, set with k_traverse's by dh_kernels_init)
hipError_t dh_region_init() {
    return hipFuncSetAttribute((const void *)k_region<1>, hipFuncAttributeMaxDynamicSharedMemorySize, RRG3 * (int)sizeof(uint32_t));
}

hipError_t dh_launch_region(const ClusterArgs &a, hipStream_t s) {
    if (a.n_frames == 0 || a.iterations == 0 || !a.pre_region || a.pre_slices < 1) return hipSuccess;
    if (a.n_frames > 65535) return hipErrorInvalidConfiguration;
    // (the rotation instance holds a CU's LDS alone; half as many, twice as long shares -- one round on the chip at 32 frames -- take
    // 0.23 instead of 0.12 ms: the time goes with the length of a share; four of a record's cells in flight change nothing)
    hipLaunchKernelGGL(k_region<1>, dim3(a.pre_slices, a.n_frames), dim3(CL_THREADS), RRG3 * sizeof(uint32_t), s, a);
    hipLaunchKernelGGL(k_region<0>, dim3(a.pre_slices, a.n_frames), dim3(CL_THREADS), 0, s, a);
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
                    a.out[k * 4 + 2] = (int32_t)((b >> 16) & 255u); a.out[k * 4 + 3] = (int32_t)(v * (b >> 24));
                }
            }
        }
    }
}

hipError_t dh_launch_votes_dump(const VotesDumpArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_votes_dump, dim3(256), dim3(256), 0, s, a);
    return hipGetLastError();
}
