// Micro-benchmark (GPU box): at what shader clock does a small, latency-bound kernel run -- right after an idle gap, back to
// back with its like, and in the middle of a sustained load?  One workgroup of 64 lanes walks a dependent LDS chain for a fixed
// number of steps and reads both counters at its start and end: clock64() (s_memtime: shader clock) and wall_clock64()
// (s_memrealtime: constant 100 MHz).  MHz = 100 x d(clock64) / d(wall_clock64); ns per dependent LDS read = wall time / steps.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/clock_probe tools/ubench/clock_probe.hip && tools/ubench/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_probe(int steps, unsigned long long *out) {
    __shared__ uint32_t ring[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) ring[i] = (uint32_t)((i * 37 + 11) & 1023);
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    uint32_t j = threadIdx.x;
    for (int s = 0; s < steps; ++s) j = ring[j];
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = j; }
}

// something that keeps every CU busy for ~ms
__global__ void __launch_bounds__(256) k_load(float *p, int n, int iters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = p[i % n];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    p[i % n] = v;
}

static void probe(const char *what, unsigned long long *d_out, int steps) {
    unsigned long long h[3];
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, steps, d_out);
    CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-44s shader clock %7.1f MHz   %6.1f ns per dependent LDS read   (%d steps in %.1f us)\n", what, 100.0 * (double)h[0] / (double)h[1],
           10.0 * (double)h[1] / steps, steps, (double)h[1] / 100.0);
}

int main() {
    unsigned long long *d_out; float *d_buf;
    CHECK(hipMalloc(&d_out, 64)); CHECK(hipMalloc(&d_buf, 1 << 24));
    CHECK(hipMemset(d_buf, 0, 1 << 24));
    const int steps = 2000;
    probe("first launch", d_out, steps);
    usleep(20000); probe("after 20 ms idle", d_out, steps);
    usleep(1000); probe("after 1 ms idle", d_out, steps);
    for (int i = 0; i < 5; ++i) probe("back to back (host sync between)", d_out, steps);
    for (int ms = 1; ms <= 64; ms *= 4) {
        // sustained load for ~ms milliseconds, then the probe right behind it on the same stream
        for (int k = 0; k < ms * 10; ++k) hipLaunchKernelGGL(k_load, dim3(4096), dim3(256), 0, 0, d_buf, 1 << 22, 2000);
        char what[64]; snprintf(what, sizeof(what), "right behind %d x k_load", ms * 10);
        probe(what, d_out, steps);
    }
    usleep(200); probe("200 us after the load", d_out, steps);
    usleep(2000); probe("2 ms after the load", d_out, steps);
    return 0;
}
