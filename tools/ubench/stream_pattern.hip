// Micro-benchmark (GPU box): does the ORDER in which k_boxsum's waves walk the frames cost memory bandwidth?  Reads 256 frames of
// 640 x 480 u16 (157 MB) and optionally writes 4 bytes per pixel position that holds something (like the sparse stores) in
//   (a) k_boxsum's pattern: wave = (frame, band of 64 + 23 rows, 256-column part), 8 bytes per lane and row, rows marched one after
//       the other, RIF rows in flight; frames dealt to the XCDs like the product kernel's grid (8, blocks, frames / 8);
//   (b) the same bytes as one linear stream (grid-stride, 16 bytes per lane);
//   (c) pattern (a) with whole rows per wave (640 columns = 1280 bytes: 20 bytes per lane as 16 + 4).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/stream_pattern tools/ubench/stream_pattern.hip && tools/ubench/stream_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define W 640
#define H 480
#define NF 256

template <int RIF, bool STORE>
__global__ void __launch_bounds__(256) k_bands(const uint16_t *frames, uint32_t *out, int band_rows, int parts, int bands, uint32_t *sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int frame = (int)blockIdx.z * 8 + (int)blockIdx.x;
    const int unit = (int)blockIdx.y * 4 + wv;
    if (frame >= NF || unit >= parts * bands) return;
    const int band = unit / parts, part = unit - band * parts;
    const int x = part * 232 + 4 * lane;
    const int y0 = band * band_rows, y1 = min(H, y0 + band_rows + 23);
    const uint16_t *img = frames + (size_t)frame * W * H;
    const bool in = x + 3 < W;
    uint32_t acc = 0;
    for (int y = y0; y < y1; y += RIF) {
        uint2 v[RIF];
#pragma unroll
        for (int k = 0; k < RIF; ++k) { const int yy = min(y + k, y1 - 1); v[k] = in ? *(const uint2 *)(img + (size_t)yy * W + x) : make_uint2(0u, 0u); }
#pragma unroll
        for (int k = 0; k < RIF; ++k) {
            acc += v[k].x ^ v[k].y;
            if (STORE && y + k < y1 && y + k >= y0 + 23 && in && (v[k].x | v[k].y)) {
                uint32_t *o = out + ((size_t)frame * H + (y + k)) * W + x;     // 16 bytes per lane where the frame holds something
                *(uint4 *)o = make_uint4(v[k].x, v[k].y, acc, 1u);
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <bool STORE>
__global__ void __launch_bounds__(256) k_linear(const uint4 *frames, uint4 *out, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = frames[i];                                             // 8 pixels
        acc += v.x ^ v.y ^ v.z ^ v.w;
        if (STORE && (v.x | v.y | v.z | v.w)) { out[2 * i] = v; out[2 * i + 1] = v; }   // 32 bytes per 8 pixels that hold something
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t npx = (size_t)NF * W * H;
    uint16_t *h = (uint16_t *)malloc(npx * 2);
    // BIWI-like occupancy: a blob of non-zero pixels in the middle third of every frame
    for (int f = 0; f < NF; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                h[((size_t)f * H + y) * W + x] = (x > 200 + (f % 60) && x < 420 + (f % 60) && y > 100 && y < 480) ? (uint16_t)(900 + ((x * 7 + y * 13 + f) & 63)) : 0;
    uint16_t *d; uint32_t *o, *sink;
    CHECK(hipMalloc(&d, npx * 2)); CHECK(hipMalloc(&o, npx * 4)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemcpy(d, h, npx * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(o, 0, npx * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time = [&](const char *what, auto launch, double bytes) {
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-72s %7.4f ms  %6.2f TB/s of pixel reads\n", what, ms, bytes / ms / 1e9);
    };
    const double rb = (double)npx * 2;
    for (int band : {64, 128}) {
        const int parts = 3, bands = (H - 23 + band - 1) / band;
        const dim3 grid(8, (parts * bands + 3) / 4, NF / 8);
        char what[128];
        snprintf(what, sizeof(what), "bands of %d rows, 256-column parts, 4 rows in flight, reads only", band);
        time(what, [&] { hipLaunchKernelGGL((k_bands<4, false>), grid, dim3(256), 0, 0, d, o, band, parts, bands, sink); }, rb);
        snprintf(what, sizeof(what), "bands of %d rows, 256-column parts, 8 rows in flight, reads only", band);
        time(what, [&] { hipLaunchKernelGGL((k_bands<8, false>), grid, dim3(256), 0, 0, d, o, band, parts, bands, sink); }, rb);
        snprintf(what, sizeof(what), "bands of %d rows, 256-column parts, 4 rows in flight, + sparse 16-byte stores", band);
        time(what, [&] { hipLaunchKernelGGL((k_bands<4, true>), grid, dim3(256), 0, 0, d, o, band, parts, bands, sink); }, rb);
    }
    time("linear stream, 16 bytes per lane, reads only", [&] { hipLaunchKernelGGL((k_linear<false>), dim3(4096), dim3(256), 0, 0, (const uint4 *)d, (uint4 *)o, npx / 8, sink); }, rb);
    time("linear stream, + sparse 32-byte stores", [&] { hipLaunchKernelGGL((k_linear<true>), dim3(4096), dim3(256), 0, 0, (const uint4 *)d, (uint4 *)o, npx / 8, sink); }, rb);
    return 0;
}
