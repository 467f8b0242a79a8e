// Issue cost of single fp64 / integer / compare / select VALU instructions for ONE wave on gfx950
// (64 back-to-back copies in a loop, inline asm so nothing is folded): what a lone wave pays per
// instruction, dependent or not.  clock64() ticks.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP64(body) ".rept 64\n" body "\n.endr"

template <int OP>
__global__ void probe(double a, double b, int iters, long long* cyc, double* sink)
{
    double x = 1.0 + 1e-3 * threadIdx.x, y = 2.0 + 1e-3 * threadIdx.x, w = a, v = b;
    int i = threadIdx.x, j = 3;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) asm volatile(REP64("v_fma_f64 %0, %0, %1, %2") : "+v"(x) : "v"(w), "v"(v));
        if (OP == 1) asm volatile(REP64("v_fma_f64 %0, %1, %2, %3") : "=v"(y) : "v"(x), "v"(w), "v"(v));
        if (OP == 2) asm volatile(REP64("v_mul_f64 %0, %0, %1") : "+v"(x) : "v"(w));
        if (OP == 3) asm volatile(REP64("v_mul_f64 %0, %1, %2") : "=v"(y) : "v"(x), "v"(w));
        if (OP == 4) asm volatile(REP64("v_add_f64 %0, %0, %1") : "+v"(x) : "v"(v));
        if (OP == 5) asm volatile(REP64("v_add_f64 %0, %1, %2") : "=v"(y) : "v"(x), "v"(v));
        if (OP == 6) asm volatile(REP64("v_mul_f64 %0, %0, %1") : "+v"(x) : "s"(a));
        if (OP == 7) asm volatile(REP64("v_fmac_f64_e32 %0, %1, %2") : "+v"(x) : "v"(w), "v"(v));
        if (OP == 8) asm volatile(REP64("v_cmp_gt_f64 vcc, %0, %1") : : "v"(x), "v"(w) : "vcc");
        if (OP == 9) asm volatile(REP64("v_cndmask_b32 %0, %1, %2, vcc") : "=v"(j) : "v"(i), "v"(i) : "vcc");
        if (OP == 10) asm volatile(REP64("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %3, %3, vcc") : "=v"(j) : "v"(x), "v"(w), "v"(i) : "vcc");
        if (OP == 11) asm volatile(REP64("v_cmp_gt_f64 s[20:21], %1, %2\n v_cndmask_b32 %0, %3, %3, s[20:21]") : "=v"(j) : "v"(x), "v"(w), "v"(i) : "s20", "s21");
        if (OP == 12) asm volatile(REP64("v_max_f64 %0, %0, %1") : "+v"(x) : "v"(w));
        if (OP == 13) asm volatile(REP64("v_cvt_i32_f64 %0, %1") : "=v"(j) : "v"(x));
        if (OP == 14) asm volatile(REP64("v_cvt_f64_i32 %0, %1") : "=v"(y) : "v"(i));
        if (OP == 15) asm volatile(REP64("v_ceil_f64 %0, %1") : "=v"(y) : "v"(x));
        if (OP == 16) asm volatile(REP64("v_add_u32 %0, %0, %1") : "+v"(j) : "v"(i));
        if (OP == 17) asm volatile(REP64("v_max_i32 %0, %0, %1") : "+v"(j) : "v"(i));
        if (OP == 18) asm volatile(REP64("v_mov_b32 %0, %1") : "=v"(j) : "v"(i));
        if (OP == 19) asm volatile(REP64("v_rcp_f64 %0, %1") : "=v"(y) : "v"(x));
        if (OP == 20) asm volatile(REP64("v_cmp_gt_f64 vcc, %0, %1\n v_mul_f64 %2, %0, %1") : : "v"(x), "v"(w), "v"(y) : "vcc");
        if (OP == 21) asm volatile(REP64("v_cmp_lt_i32 vcc, %0, %1") : : "v"(i), "v"(j) : "vcc");
        if (OP == 22) asm volatile(REP64("s_and_b64 s[20:21], s[20:21], exec") : : : "s20", "s21", "scc");
        if (OP == 23) asm volatile(REP64("v_mul_f64 %0, %1, %2\n s_and_b64 s[20:21], s[20:21], exec") : "=v"(y) : "v"(x), "v"(w) : "s20", "s21", "scc");
        if (OP == 24) asm volatile(REP64("v_mul_f64 %0, %2, %3\n v_mov_b32 %1, %4") : "=v"(y), "=v"(j) : "v"(x), "v"(w), "v"(i));
    }
    long long t1 = clock64();
    sink[threadIdx.x] = x + y + j;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
static void run(const char* name, int per, long long* c, double* sink)
{
    const int iters = 2000;
    long long best = 1LL << 62;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((probe<OP>), dim3(1), dim3(64), 0, 0, 0.999999, 1e-7, iters, c, sink);
        (void)hipDeviceSynchronize();
        long long cy; (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        if (cy < best) best = cy;
    }
    printf("%-44s %6.2f ticks per group of %d\n", name, (double)best / (iters * 64.0), per);
}

int main()
{
    long long* c; double* sink;
    (void)hipMalloc(&c, 8); (void)hipMalloc(&sink, 64 * 8);
    run<0>("v_fma_f64 dependent", 1, c, sink);
    run<1>("v_fma_f64 independent", 1, c, sink);
    run<2>("v_mul_f64 dependent", 1, c, sink);
    run<3>("v_mul_f64 independent", 1, c, sink);
    run<4>("v_add_f64 dependent", 1, c, sink);
    run<5>("v_add_f64 independent", 1, c, sink);
    run<6>("v_mul_f64 dependent, SGPR operand", 1, c, sink);
    run<7>("v_fmac_f64_e32 dependent", 1, c, sink);
    run<8>("v_cmp_gt_f64 -> vcc", 1, c, sink);
    run<9>("v_cndmask_b32 (vcc)", 1, c, sink);
    run<10>("v_cmp_gt_f64 vcc ; v_cndmask vcc", 2, c, sink);
    run<11>("v_cmp_gt_f64 sgpr ; v_cndmask sgpr", 2, c, sink);
    run<12>("v_max_f64 dependent", 1, c, sink);
    run<13>("v_cvt_i32_f64", 1, c, sink);
    run<14>("v_cvt_f64_i32", 1, c, sink);
    run<15>("v_ceil_f64", 1, c, sink);
    run<16>("v_add_u32 dependent", 1, c, sink);
    run<17>("v_max_i32 dependent", 1, c, sink);
    run<18>("v_mov_b32", 1, c, sink);
    run<19>("v_rcp_f64", 1, c, sink);
    run<20>("v_cmp_gt_f64 vcc ; v_mul_f64", 2, c, sink);
    run<21>("v_cmp_lt_i32 -> vcc", 1, c, sink);
    run<22>("s_and_b64", 1, c, sink);
    run<23>("v_mul_f64 ; s_and_b64", 2, c, sink);
    run<24>("v_mul_f64 ; v_mov_b32", 2, c, sink);
    return 0;
}
