// Micro-benchmark (GPU box): is the 16-byte node gather of k_traverse's walks bound by the vector-memory path's throughput or
// by its latency?  Dependent 16-byte gathers from a 2600-node table; 512 workgroups (2 per CU) of 256 / 512 / 1024 threads
// (8 / 16 / 32 waves per CU), W independent chains per lane, G lanes sharing an index.  If the cost per wave-instruction per CU
// stays put as the occupancy falls, the path's throughput bounds it; if it grows in proportion, its latency does.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W>
__global__ void __launch_bounds__(1024) k_chase(const uint4 *tab, int n, int iters, int G, uint32_t *out) {
    uint32_t idx[W], acc = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) idx[i] = (((blockIdx.x * 1024 + threadIdx.x) / G + i * 977) * 2654435761u) % (uint32_t)n;
    for (int it = 0; it < iters; ++it) {
        uint4 v[W];
#pragma unroll
        for (int i = 0; i < W; ++i) v[i] = tab[idx[i]];
#pragma unroll
        for (int i = 0; i < W; ++i) { acc += v[i].y ^ v[i].z; idx[i] = v[i].x + (v[i].w & 1); }
    }
    uint32_t s = acc;
#pragma unroll
    for (int i = 0; i < W; ++i) s += idx[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2600, iters = 2000, blocks = 512;
    const bool quick = argc > 2;
    std::vector<uint4> h(n);
    for (int i = 0; i < n; ++i) h[i] = make_uint4((uint32_t)((i * 7919ull + 13) % n), i, i * 3, 0);
    uint4 *d; uint32_t *out;
    CHECK(hipMalloc(&d, n * 16)); CHECK(hipMalloc(&out, blocks * 1024 * 4));
    CHECK(hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("16-byte dependent gathers, table of %d nodes; ns per wave-instruction per CU, and the latency of one gather that would explain it\n", n);
    for (int G : {1, 4, 64})
        for (int threads : {256, 512, 1024})
            for (int W : {1, 2, 3}) {
                if (quick && (threads != 1024 || W != 3)) continue;
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0));
                    if (W == 1) hipLaunchKernelGGL(k_chase<1>, dim3(blocks), dim3(threads), 0, 0, d, n, iters, G, out);
                    else if (W == 2) hipLaunchKernelGGL(k_chase<2>, dim3(blocks), dim3(threads), 0, 0, d, n, iters, G, out);
                    else hipLaunchKernelGGL(k_chase<3>, dim3(blocks), dim3(threads), 0, 0, d, n, iters, G, out);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                const float waves = 2.0f * threads / 64.0f;
                printf("G=%2d  %2.0f waves/CU  W=%d  %8.3f ms  %7.2f ns/wave-instr/CU   (round trip %6.1f ns)\n", G, waves, W, best,
                       best * 1e6f / (waves * iters * W), best * 1e6f / iters);
            }
    return 0;
}
