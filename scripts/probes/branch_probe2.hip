// Taken skip-branch cost versus the size of the skipped block, ONE in-order wave on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP(x, a) x = x * a + 1.0;
#define CHAIN10(x, a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a) OP(x,a)

template <int NRARE>  // number of mul+add pairs in the (never executed) rare block; 0 = no branch at all
__global__ void probe(double* out, double thr, int iters, long long* cyc)
{
    double x = out[threadIdx.x], a = 0.999999;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int b = 0; b < 10; b++) {
            CHAIN10(x, a)
            if (NRARE > 0) {
                if (x > thr) {  // never true
#pragma unroll
                    for (int k = 0; k < NRARE; k++) { OP(x, a) }
                    x = sqrt(x);
                }
            }
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int N>
static void run(double* d, long long* c, const double* h, int iters, long long base)
{
    long long best = 1LL << 62;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipMemcpy(d, h, 64 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe<N>, dim3(1), dim3(64), 0, 0, d, 1e300, iters, c);
        (void)hipDeviceSynchronize();
        long long cy; (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        if (cy < best) best = cy;
    }
    printf("rare block of %3d op pairs: %.1f cycles per 20-instr block, branch cost %.1f cycles\n", N,
           (double)best / iters / 10, base ? ((double)best - base) / iters / 10 : 0.0);
}

int main()
{
    double* d; long long* c;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 8);
    double h[64]; for (int i = 0; i < 64; i++) h[i] = 1.0 + i * 1e-3;
    int iters = 20000;
    // baseline
    long long best = 1LL << 62;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipMemcpy(d, h, 64 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d, 1e300, iters, c);
        (void)hipDeviceSynchronize();
        long long cy; (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        if (cy < best) best = cy;
    }
    printf("baseline: %.1f cycles per 20-instr block\n", (double)best / iters / 10);
    run<1>(d, c, h, iters, best); run<3>(d, c, h, iters, best); run<6>(d, c, h, iters, best);
    run<8>(d, c, h, iters, best); run<12>(d, c, h, iters, best); run<20>(d, c, h, iters, best);
    run<40>(d, c, h, iters, best); run<80>(d, c, h, iters, best); run<160>(d, c, h, iters, best);
    return 0;
}
