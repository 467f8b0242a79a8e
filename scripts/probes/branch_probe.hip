// Cost of a taken skip-branch (s_cbranch_execz over a rare block) for ONE in-order wave on gfx950.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o branch_probe branch_probe.hip && ./branch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHAIN20(x, a) \
    x = x * a + 1.0; x = x * a + 1.0; x = x * a + 1.0; x = x * a + 1.0; x = x * a + 1.0; \
    x = x * a + 1.0; x = x * a + 1.0; x = x * a + 1.0; x = x * a + 1.0; x = x * a + 1.0;

template <int MODE>  // 0: no branches, 1: rare block behind a lane-varying test (never taken), 2: same, wave-uniform test
__global__ void probe(double* out, double thr, int iters, long long* cyc)
{
    double x = out[threadIdx.x], a = 0.999999;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int b = 0; b < 10; b++) {
            CHAIN20(x, a)
            if (MODE == 1) {
                if (x > thr) {  // never true
#pragma unroll
                    for (int k = 0; k < 6; k++) { CHAIN20(x, a) }
                    x = sqrt(x);
                }
            } else if (MODE == 2) {
                if (__any(x > thr)) {
#pragma unroll
                    for (int k = 0; k < 6; k++) { CHAIN20(x, a) }
                    x = sqrt(x);
                }
            }
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    double* d; long long* c;
    hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
    double h[64]; for (int i = 0; i < 64; i++) h[i] = 1.0 + i * 1e-3;
    int iters = 20000;
    for (int mode = 0; mode < 3; mode++) {
        long long best = 1LL << 62;
        for (int rep = 0; rep < 3; rep++) {
            hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d, 1e300, iters, c);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d, 1e300, iters, c);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, d, 1e300, iters, c);
            hipDeviceSynchronize();
            long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            if (cy < best) best = cy;
        }
        // clock64 = s_memtime: 100 MHz constant clock on gfx9?  report raw and per block
        printf("mode %d: %lld ticks, %.3f ticks per block of 20 dependent fp64 ops (+branch)\n", mode, best,
               (double)best / iters / 10);
    }
    return 0;
}
