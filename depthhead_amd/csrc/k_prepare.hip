// k_prepare.hip -- per batch, ahead of the walks: rectangle-sum images and tile flags (k_boxsum, k_pixflags), tile lists (k_tile_list), zero-fill (k_zero)
//
// One of the kernel translation units of libdepthhead_hip.so (hand-written HIP for gfx950: wave64, 160 KB LDS/CU;
// no MFMA anywhere -- there is no dense contraction on this path).  Overview of the pipeline: dh_api.hip.
#include "dh_device.h"

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
    const uint8_t tag = (uint8_t)*a.gen;             // this batch's tile-flag tag
    const float r_tpx = 1.0f / (float)a.tpx, r_tpy = 1.0f / (float)a.tpy;
    uint2 e[RIF], l[RIF], en[RIF], ln[RIF];
    int rs = 0;                          // ring slot of step t: t mod (rh - 1)
    // sparse stores (BoxArgs::blk_mask): the previous fill's non-zero mask of the 32-row block being emitted, and the next block's
    unsigned long long *bm = a.blk_mask ? a.blk_mask + ((size_t)frame * a.mask_blocks) * a.parts + part : nullptr;
    const int bsh = a.blk_shift, bmk = (1 << a.blk_shift) - 1;     // mask blocks of 32 rows (8 on single-frame workspaces)
    unsigned long long pblk = bm ? bm[(size_t)(Y0 >> bsh) * a.parts] : ~0ull;
    unsigned long long pnext = bm && ((Y0 >> bsh) + 1) < a.mask_blocks ? bm[(size_t)((Y0 >> bsh) + 1) * a.parts] : ~0ull;
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
            // Which k_traverse tiles have a non-zero rectangle sum in their region?  Every mask block of output rows
            // (and at the end of the band) each lane that saw a non-zero sum marks the tiles whose regions
            // contain its columns and those rows (plain stores of the batch's tag).
            const int yo = Y0 + t - warm;
            if ((yo & bmk) == bmk || yo == y_end - 1) {
                if (bm) {
                    const unsigned long long nm = __ballot(acc != 0);
                    if (lane == 0) bm[(size_t)(yo >> bsh) * a.parts] = nm;
                    pblk = pnext;
                    pnext = ((yo >> bsh) + 2) < a.mask_blocks ? bm[(size_t)((yo >> bsh) + 2) * a.parts] : ~0ull;
                }
                if (acc != 0) {
                    // tile tx covers columns [tx * tpx, tx * tpx + tbw): tx in [(x + 3 - tbw) / tpx + 1 .. x / tpx] clipped
                    const int y_lo = max(yo & ~bmk, Y0);
                    const int tx1 = min(div_small(x + 3, a.tpx, r_tpx), a.tiles_x - 1), tx0 = max(x - a.tbw < 0 ? 0 : div_small(x - a.tbw, a.tpx, r_tpx) + 1, 0);
                    const int ty1 = min(div_small(yo, a.tpy, r_tpy), a.tiles_y - 1), ty0 = max(y_lo - a.tbh + 1 <= 0 ? 0 : div_small(y_lo - a.tbh, a.tpy, r_tpy) + 1, 0);
                    for (int ty = ty0; ty <= ty1; ++ty)
                        for (int tx = tx0; tx <= tx1; ++tx) flags[ty * a.tiles_x + tx] = tag;
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
    if (a.zero_ptr) {
        // the batch's other counters (hit counters, guess grids, window counts, leaf histogram): this is the first kernel of the
        // batch and none of them is touched before the next one, so every wave of the grid clears a share -- no fill dispatch
        const size_t nw = (size_t)gridDim.x * gridDim.y * gridDim.z * BOXW_THREADS;
        const size_t gi = (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * BOXW_THREADS + threadIdx.x;
        for (size_t i = gi; i < a.zero_lo; i += nw) a.zero_ptr[i] = 0u;
        for (size_t i = (size_t)a.zero_hi + gi; i < a.zero_end; i += nw) a.zero_ptr[i] = 0u;
    }
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
    const dim3 grid((unsigned)std::min(8, a.n_frames), a.blocks_per_frame, fb), block(BOXW_THREADS);
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

// ================================================================== k_pixflags
// General (mixed-rectangle) path: which k_traverse tiles have a non-zero pixel under their footprint?
// One workgroup scans a band of 32 image rows of one frame (8-byte loads, 4 columns per lane); a lane
// that saw a non-zero pixel marks the tiles whose footprints contain its columns and the band's rows
// (plain stores of the batch's tag, BoxArgs::gen).  Conservative by construction.
__global__ void __launch_bounds__(256) k_pixflags(PixFlagArgs a) {
    const int frame = (int)blockIdx.z * 8 + (int)blockIdx.x;
    if (frame >= a.n_frames) return;
    const int y0 = (int)blockIdx.y * 32, y1 = min(y0 + 32, a.h);
    const uint16_t *img = a.frames + (size_t)frame * a.w * a.h;
    uint8_t *flags = a.tile_flags + (size_t)frame * a.tiles_x * a.tiles_y;
    const uint8_t tag = (uint8_t)*a.gen;
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
            for (int tx = tx0; tx <= tx1; ++tx) flags[ty * a.tiles_x + tx] = tag;
    }
}

hipError_t dh_launch_pixflags(const PixFlagArgs &a, hipStream_t s) {
    const int fb = (a.n_frames + 7) / 8, bands = (a.h + 31) / 32;
    if (fb == 0 || bands == 0) return hipSuccess;
    if (bands > 65535 || fb > 65535) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_pixflags, dim3(8, bands, fb), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ================================================================== k_tile_list
// Which (frame, tile) pairs does k_traverse have to work on?  One list per x = frame mod 8 (the grid's x, which picks the XCD),
// in the order the grid visits them (frame group, then tile), entry = group << 16 | tile.  k_traverse's workgroup (x, k) takes the
// k-th entry; the workgroups beyond a list's end -- the flagged-empty tiles, four in seven on the bench frames -- then sit at the
// END of the grid, where they leave at once instead of each holding a workgroup slot (79 KB of LDS) for the 1-2 us it takes a
// workgroup to start, read its flag and go, in the middle of the real work.
__global__ void __launch_bounds__(1024) k_tile_list(const uint8_t *flags, const uint32_t *gen, int n_frames, int tiles, uint32_t *list, uint32_t *count, uint32_t stride) {
    __shared__ uint32_t wsum[16];
    const uint8_t tag = (uint8_t)*gen;
    const int x = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    const uint32_t fb = (uint32_t)(n_frames + 7) / 8, total = fb * (uint32_t)tiles;
    uint32_t base = 0;
    for (uint32_t e0 = 0; e0 < total; e0 += 1024) {
        const uint32_t e = e0 + (uint32_t)tid;
        const uint32_t z = e / (uint32_t)tiles, t = e - z * (uint32_t)tiles;
        const uint32_t frame = z * 8 + (uint32_t)x;
        const bool on = e < total && frame < (uint32_t)n_frames && flags[(size_t)frame * tiles + t] == tag;
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

hipError_t dh_launch_tile_list(const uint8_t *flags, const uint32_t *gen, int n_frames, int tiles, uint32_t *list, uint32_t *count, uint32_t stride, hipStream_t s) {
    if (n_frames == 0 || tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tile_list, dim3(8), dim3(1024), 0, s, flags, gen, n_frames, tiles, list, count, stride);
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
