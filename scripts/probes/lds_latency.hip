// Dependent LDS read latency for ONE wave on gfx950: pointer chase through a 96 KB table with
// ds_read_b128 (the kernel's {c, cp} pair reads) and ds_read_b64.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>  // 0: two b128 reads per hop (entries j, j+1), 1: one b64 read per hop, 2: one b128
__global__ void chase(const int* init, int n, int iters, long long* cyc, int* sink)
{
    extern __shared__ double2 tab[];
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        int nxt = init[j];
        tab[j] = make_double2(__int_as_float(0) + (double)nxt, (double)(nxt ^ 1));
    }
    __syncthreads();
    int j = (threadIdx.x * 97) % (n - 1);
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            double2 a = tab[j], b = tab[j + 1];
            j = (int)(a.x + 0.0 * b.y);
        } else if (MODE == 1) {
            j = (int)((const double*)tab)[2 * j];
        } else {
            double2 a = tab[j];
            j = (int)(a.x + 0.0 * a.y);
        }
        j = min(max(j, 0), n - 2);
    }
    long long t1 = clock64();
    sink[threadIdx.x] = j;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    const int n = 6000, iters = 20000;
    int* h = new int[n];
    unsigned s = 12345;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = (int)(s % (unsigned)(n - 1)); }
    int *d, *sink; long long* c;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&sink, 64 * 4); (void)hipMalloc(&c, 8);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; mode++) {
        long long best = 1LL << 62;
        for (int rep = 0; rep < 3; rep++) {
            if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(1), dim3(64), n * 16, 0, d, n, iters, c, sink);
            if (mode == 1) hipLaunchKernelGGL(chase<1>, dim3(1), dim3(64), n * 16, 0, d, n, iters, c, sink);
            if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(1), dim3(64), n * 16, 0, d, n, iters, c, sink);
            (void)hipDeviceSynchronize();
            long long cy; (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            if (cy < best) best = cy;
        }
        printf("mode %d: %.1f cycles per dependent hop (incl. ~6 VALU of index arithmetic)\n", mode, (double)best / iters);
    }
    return 0;
}
