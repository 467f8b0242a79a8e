// What does the shader clock do under load?  Block 0 times a fixed chain of dependent fp64 FMAs with both counters
// (s_memtime: shader clock; s_memrealtime: the constant 100 MHz reference) while `grid - 1` other single-wave
// workgroups run the same chain: shader MHz = d(memtime) / d(memrealtime) * 100.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out, int iters, long long* rec)
{
    double x = out[threadIdx.x & 63], a = 0.999999;
    const long long c0 = __builtin_readcyclecounter();
    const long long w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 32; k++) x = x * a + 1.0;
    }
    const long long c1 = __builtin_readcyclecounter();
    const long long w1 = __builtin_amdgcn_s_memrealtime();
    if (x == 12345.678) out[0] = x;
    if (threadIdx.x == 0) { rec[2 * blockIdx.x] = c1 - c0; rec[2 * blockIdx.x + 1] = w1 - w0; }
}
int main()
{
    double* d; long long* r;
    const int maxg = 4096;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&r, maxg * 16);
    (void)hipMemset(d, 0, 64 * 8);
    const int iters = 60000;   // 1.92 M dependent FMAs: ~3.5 ms
    std::vector<long long> h(maxg * 2);
    for (int rep = 0; rep < 2; rep++)
    for (int grid : {1, 2, 16, 256, 512, 1024, 2048, 4096}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, d, iters, r);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), r, grid * 16, hipMemcpyDeviceToHost);
        double mhz_min = 1e9, mhz_max = 0, cyc = 0;
        for (int b = 0; b < grid; b++) {
            const double mhz = (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;
            mhz_min = mhz < mhz_min ? mhz : mhz_min; mhz_max = mhz > mhz_max ? mhz : mhz_max;
            cyc += (double)h[2 * b];
        }
        printf("waves %4d: kernel %.3f ms; block 0: %.0f shader cycles in %.3f ms = %.0f MHz (all blocks %.0f..%.0f MHz), %.2f cycles per FMA\n",
               grid, ms, (double)h[0], (double)h[1] / 1e5, (double)h[0] / (double)h[1] * 100.0, mhz_min, mhz_max, cyc / grid / ((double)iters * 32));
    }
    return 0;
}
