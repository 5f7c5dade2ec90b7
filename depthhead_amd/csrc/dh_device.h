// dh_device.h -- device-side helpers shared by the kernel translation units (k_*.hip): Rust `as` casts, the
// reference's f32 matrix-vector order, wave-level scans.  Not part of the ABI.
#pragma once
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

// Compact node of the uniform-rectangle path (built by k_nodes_compact in k_forest.hip, walked by k_traverse).
struct __attribute__((aligned(16))) NodeU {
    uint32_t offs;      // LDS offset of r1's box sum | r2's << 14 | amb << 28, relative to the patch origin
    int32_t  ilo;
    int32_t  child_zero;
    int32_t  child_one;
};

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

// Window list entries (leaf reached per (tree, active window)): 16-bit when the forest has at most 65 535 leaves -- half the
// bytes k_traverse writes and k_emit reads back -- else 32-bit.  `ls` = log2 of the entry size.
__device__ __forceinline__ void store_leaf(char *base, int ls, int idx, int32_t leaf) {
    if (ls == 1) ((uint16_t *)base)[idx] = (uint16_t)leaf;
    else ((int32_t *)base)[idx] = leaf;
}
__device__ __forceinline__ uint32_t load_leaf(const char *base, int ls, size_t idx) {
    return ls == 1 ? (uint32_t)((const uint16_t *)base)[idx] : (uint32_t)((const int32_t *)base)[idx];
}
