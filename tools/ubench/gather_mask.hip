// Micro-benchmark (GPU box): what does a 16-byte gather cost when only SOME lanes of the wave are active (exec mask)?
// k_traverse's lock-step walk loop keeps every lane active to the end of the deepest walk of its wave (finished walks re-read
// one shared entry); if the vector-memory path charges by ACTIVE lanes (or quads), masking finished walks out -- or fetching a
// node once for several adjacent windows of a lane and again only for the lanes whose windows diverged -- would cut its load.
// Dependent 16-byte gathers, 2600-node table, 512 workgroups x 1024 threads (32 waves per CU), 3 chains per lane; active lanes
// are the first A of the wave ("low"), every (64/A)-th lane ("strided") or A random lanes per wave ("random").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W>
__global__ void __launch_bounds__(1024) k_chase(const uint4 *tab, int n, int iters, int G, unsigned long long mask, int rot, uint32_t *out) {
    uint32_t idx[W], acc = 0;
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 1024 + threadIdx.x) >> 6;
    const int sh = rot ? (wave * 7) & 63 : 0;                                   // "random": another rotation of the mask per wave
    const unsigned long long m = sh ? (mask << sh) | (mask >> (64 - sh)) : mask;
#pragma unroll
    for (int i = 0; i < W; ++i) idx[i] = (((blockIdx.x * 1024 + threadIdx.x) / G + i * 977) * 2654435761u) % (uint32_t)n;
    if ((m >> lane) & 1ull) {
        for (int it = 0; it < iters; ++it) {
            uint4 v[W];
#pragma unroll
            for (int i = 0; i < W; ++i) v[i] = tab[idx[i]];
#pragma unroll
            for (int i = 0; i < W; ++i) { acc += v[i].y ^ v[i].z; idx[i] = v[i].x + (v[i].w & 1); }
        }
    }
    uint32_t s = acc;
#pragma unroll
    for (int i = 0; i < W; ++i) s += idx[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2600, iters = 2000, blocks = 512, threads = 1024;
    std::vector<uint4> h(n);
    for (int i = 0; i < n; ++i) h[i] = make_uint4((uint32_t)((i * 7919ull + 13) % n), i, i * 3, 0);
    uint4 *d; uint32_t *out;
    CHECK(hipMalloc(&d, n * 16)); CHECK(hipMalloc(&out, blocks * 1024 * 4));
    CHECK(hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("16-byte dependent gathers under an exec mask, table of %d nodes, 32 waves/CU, 3 chains per lane: ns per wave-instruction per CU\n", n);
    for (int G : {1, 4})
        for (int A : {64, 48, 32, 16, 8, 4, 1})
            for (int pat = 0; pat < 3; ++pat) {
                unsigned long long mask = 0;
                if (pat == 0) mask = A == 64 ? ~0ull : (1ull << A) - 1ull;
                else if (pat == 1) { if (64 % A) continue; for (int i = 0; i < 64; i += 64 / A) mask |= 1ull << i; }
                else { unsigned s = 12345; int c = 0; while (c < A) { s = s * 1664525u + 1013904223u; int b = (s >> 24) & 63; if (!((mask >> b) & 1ull)) { mask |= 1ull << b; ++c; } } }
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_chase<3>, dim3(blocks), dim3(threads), 0, 0, d, n, iters, G, mask, pat == 2, out);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                const float waves = 2.0f * threads / 64.0f;
                printf("G=%d  active %2d %-8s %8.3f ms  %7.2f ns/wave-instr/CU\n", G, A, pat == 0 ? "low" : pat == 1 ? "strided" : "random", best, best * 1e6f / (waves * iters * 3));
            }
    return 0;
}
