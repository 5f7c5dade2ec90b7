// Micro-benchmark (GPU box): cost of the node gathers of k_traverse's walk loop on gfx950.
// Every lane chases indices through a small table (L1/L2 resident), 16 waves per workgroup, 2 workgroups per CU.
//   mode 0: 16-byte gather, all lanes      mode 1: 8-byte gather       mode 2: 4-byte gather
//   mode 3: 16-byte gather, every 4th lane active      mode 4: 16-byte gather, lanes 0..15 active
//   mode 5: 4-byte gather per lane + quad exchange (each quad shares one node)
// G = lanes per group that share an index (coherence).  Prints ns per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) k_chase(const uint4 *tab16, const uint2 *tab8, const uint32_t *tab4, int n, int iters, int G, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    uint32_t idx = ((blockIdx.x * 1024 + threadIdx.x) / G * 2654435761u) % (uint32_t)n;
    uint32_t acc = 0;
    const bool on = MODE == 3 ? (lane & 3) == 0 : MODE == 4 ? lane < 16 : true;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 3 || MODE == 4) {
            if (on) { const uint4 v = tab16[idx]; acc += v.y ^ v.z; idx = v.x + (v.w & 1); }
        } else if (MODE == 1) {
            const uint2 v = tab8[idx]; acc += v.y; idx = v.x;
        } else if (MODE == 2) {
            const uint32_t v = tab4[idx]; acc += v; idx = v;
        } else if (MODE == 5) {
            // quad leader's index; lane j of the quad loads dword j of the 16-byte node, then the quad exchanges
            const uint32_t lead = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
            const uint32_t w = ((const uint32_t *)tab16)[lead * 4 + (lane & 3)];
            const uint32_t x = __builtin_amdgcn_mov_dpp(w, 0x00, 0xf, 0xf, true);          // dword 0 of the node
            const uint32_t y = __builtin_amdgcn_mov_dpp(w, 0x55, 0xf, 0xf, true);          // dword 1
            const uint32_t z = __builtin_amdgcn_mov_dpp(w, 0xaa, 0xf, 0xf, true);          // dword 2
            const uint32_t q = __builtin_amdgcn_mov_dpp(w, 0xff, 0xf, 0xf, true);          // dword 3
            acc += y ^ z; idx = x + (q & 1);
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc + idx;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2600;
    const int iters = 2000, blocks = 512;
    std::vector<uint4> h16(n); std::vector<uint2> h8(n); std::vector<uint32_t> h4(n);
    for (int i = 0; i < n; ++i) {
        const uint32_t nx = (uint32_t)((i * 7919ull + 13) % n);
        h16[i] = make_uint4(nx, i, i * 3, 0); h8[i] = make_uint2(nx, i); h4[i] = nx;
    }
    uint4 *d16; uint2 *d8; uint32_t *d4, *out;
    CHECK(hipMalloc(&d16, n * 16)); CHECK(hipMalloc(&d8, n * 8)); CHECK(hipMalloc(&d4, n * 4)); CHECK(hipMalloc(&out, blocks * 1024 * 4));
    CHECK(hipMemcpy(d16, h16.data(), n * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d8, h8.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d4, h4.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("table of %d nodes; ns per wave-instruction per CU (2 x 16 waves per CU, dependent chains)\n", n);
    const char *names[] = {"16B all lanes", "8B all lanes", "4B all lanes", "16B every 4th lane", "16B lanes 0..15", "4B + quad exchange"};
    for (int G : {1, 4, 16, 64}) {
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                switch (mode) {
                case 0: hipLaunchKernelGGL(k_chase<0>, dim3(blocks), dim3(1024), 0, 0, d16, d8, d4, n, iters, G, out); break;
                case 1: hipLaunchKernelGGL(k_chase<1>, dim3(blocks), dim3(1024), 0, 0, d16, d8, d4, n, iters, G, out); break;
                case 2: hipLaunchKernelGGL(k_chase<2>, dim3(blocks), dim3(1024), 0, 0, d16, d8, d4, n, iters, G, out); break;
                case 3: hipLaunchKernelGGL(k_chase<3>, dim3(blocks), dim3(1024), 0, 0, d16, d8, d4, n, iters, G, out); break;
                case 4: hipLaunchKernelGGL(k_chase<4>, dim3(blocks), dim3(1024), 0, 0, d16, d8, d4, n, iters, G, out); break;
                case 5: hipLaunchKernelGGL(k_chase<5>, dim3(blocks), dim3(1024), 0, 0, d16, d8, d4, n, iters, G, out); break;
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            // wave-instructions per CU: 32 waves x iters
            printf("G=%2d  %-22s %8.3f ms   %7.2f ns/wave-instr/CU\n", G, names[mode], best, best * 1e6f / (32.0f * iters));
        }
    }
    return 0;
}
