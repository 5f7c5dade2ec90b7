// k_aux.hip -- sibling consumers and ingest: predict_mask, 2-D Hough variant (votes, blur, argmax), BIWI run-length decode
//
// One of the kernel translation units of libdepthhead_hip.so (hand-written HIP for gfx950: wave64, 160 KB LDS/CU;
// no MFMA anywhere -- there is no dense contraction on this path).  Overview of the pipeline: dh_api.hip.
#include "dh_device.h"

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
