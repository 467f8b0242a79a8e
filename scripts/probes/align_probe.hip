// Does an 8-byte (VOP3) instruction that straddles an instruction-fetch boundary cost a lone wave
// extra cycles?  64 v_mul_f64 (8 bytes each) per loop trip, the block starting on a 64-byte
// boundary (+0) or 4 bytes after one (+4: every 4th instruction straddles a 32-byte boundary), and
// mixes of 8- and 4-byte instructions.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void probe(double a, int iters, long long* cyc, double* sink)
{
    double x = 1.0 + 1e-3 * threadIdx.x, w = a;
    int j = threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) asm volatile(".p2align 6\n .rept 64\n v_mul_f64 %0, %0, %1\n .endr" : "+v"(x) : "v"(w));
        if (OP == 1) asm volatile(".p2align 6\n s_nop 0\n .rept 64\n v_mul_f64 %0, %0, %1\n .endr" : "+v"(x) : "v"(w));
        if (OP == 2) asm volatile(".p2align 6\n s_nop 0\n s_nop 0\n .rept 64\n v_mul_f64 %0, %0, %1\n .endr" : "+v"(x) : "v"(w));
        // 3 x 8 bytes + 2 x 4 bytes = 32 bytes: never straddles 32 (aligned start)
        if (OP == 3) asm volatile(".p2align 6\n .rept 16\n v_mul_f64 %0, %0, %2\n v_mul_f64 %0, %0, %2\n v_mul_f64 %0, %0, %2\n v_mov_b32 %1, %1\n v_mov_b32 %1, %1\n .endr" : "+v"(x), "+v"(j) : "v"(w));
        // same mix, 4-byte instructions apart: 8 8 4 8 4 -> the third multiply straddles 16-byte, not 32
        if (OP == 4) asm volatile(".p2align 6\n .rept 16\n v_mul_f64 %0, %0, %2\n v_mul_f64 %0, %0, %2\n v_mov_b32 %1, %1\n v_mul_f64 %0, %0, %2\n v_mov_b32 %1, %1\n .endr" : "+v"(x), "+v"(j) : "v"(w));
        // 4 8 8 8 4: the three multiplies sit at +4, +12, +20 : none crosses the 32-byte boundary
        if (OP == 5) asm volatile(".p2align 6\n .rept 16\n v_mov_b32 %1, %1\n v_mul_f64 %0, %0, %2\n v_mul_f64 %0, %0, %2\n v_mul_f64 %0, %0, %2\n v_mov_b32 %1, %1\n .endr" : "+v"(x), "+v"(j) : "v"(w));
        // 8 8 8 4 | 4 ... shifted by 4: pattern 4 8 8 8 8(straddle) ...
        if (OP == 6) asm volatile(".p2align 6\n s_nop 0\n .rept 16\n v_mul_f64 %0, %0, %2\n v_mul_f64 %0, %0, %2\n v_mul_f64 %0, %0, %2\n v_mov_b32 %1, %1\n v_mov_b32 %1, %1\n .endr" : "+v"(x), "+v"(j) : "v"(w));
    }
    long long t1 = clock64();
    sink[threadIdx.x] = x + j;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
static void run(const char* name, int n, long long* c, double* sink)
{
    const int iters = 2000;
    long long best = 1LL << 62;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((probe<OP>), dim3(1), dim3(64), 0, 0, 0.999999, iters, c, sink);
        (void)hipDeviceSynchronize();
        long long cy; (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        if (cy < best) best = cy;
    }
    printf("%-58s %6.2f cycles per instruction\n", name, (double)best / (iters * (double)n));
}

int main()
{
    long long* c; double* sink;
    (void)hipMalloc(&c, 8); (void)hipMalloc(&sink, 64 * 8);
    run<0>("64 x v_mul_f64, block at 64-byte boundary", 64, c, sink);
    run<1>("64 x v_mul_f64, block at +4", 65, c, sink);
    run<2>("64 x v_mul_f64, block at +8", 66, c, sink);
    run<3>("16 x (8 8 8 4 4), aligned", 80, c, sink);
    run<4>("16 x (8 8 4 8 4), aligned", 80, c, sink);
    run<5>("16 x (4 8 8 8 4), aligned", 80, c, sink);
    run<6>("16 x (8 8 8 4 4), at +4", 81, c, sink);
    return 0;
}
