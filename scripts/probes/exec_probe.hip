// Does a wave64 fp64 VALU instruction get cheaper with fewer active lanes on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP(x, a) x = x * a + 1.0;
#define CHAIN10(x, a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a)
__global__ void probe(double* out, int active, int iters, long long* cyc)
{
    double x = out[threadIdx.x], a = 0.999999;
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < active) {
        t0 = clock64();
        for (int it = 0; it < iters; it++) { CHAIN10(x, a) CHAIN10(x, a) CHAIN10(x, a) CHAIN10(x, a) }
        t1 = clock64();
    }
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double* d; long long* c;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 8);
    double h[64]; for (int i = 0; i < 64; i++) h[i] = 1.0 + i * 1e-3;
    int iters = 20000;
    for (int active : {64, 48, 32, 17, 16, 8, 1}) {
        long long best = 1LL << 62;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, active, iters, c);
            (void)hipDeviceSynchronize();
            long long cy; (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            if (cy < best) best = cy;
        }
        printf("active lanes %2d: %.2f cycles per dependent fp64 instruction\n", active, (double)best / iters / 80);
    }
    return 0;
}
