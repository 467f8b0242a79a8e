// pgr_debug_entry.h -- host entries of the unit-level kernels (parity tests only): pgr_debug_math, pgr_debug_step, pgr_eval_points.
// (Part of the ONE translation unit pgr_hip.hip, included there in this order; not a stand-alone header.)
#ifndef PGR_DEBUG_ENTRY_H
#define PGR_DEBUG_ENTRY_H

extern "C" int pgr_debug_math(const double* a, const double* b, int64_t M, double* out9)
{
    if (!a || !b || !out9 || M <= 0) return fail("pgr_debug_math: bad argument");
    struct Buf { void* p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } da, db, dout;
    HIPCHK(hipMalloc(&da.p, M * 8));
    HIPCHK(hipMalloc(&db.p, M * 8));
    HIPCHK(hipMalloc(&dout.p, M * 72));
    HIPCHK(hipMemcpy(da.p, a, M * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, b, M * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pgr_math_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0,
                       (const double*)da.p, (const double*)db.p, M, (double*)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out9, dout.p, M * 72, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int pgr_debug_step(pgr_env* env, const double* t, const double* y, const double* h, int64_t M,
                              double rtol, double atol, double* out11)
{
    if (!env || !t || !y || !h || !out11 || M <= 0) return fail("pgr_debug_step: bad argument");
    HIPCHK(hipSetDevice(env->device));
    DevBuf dt, dy, dh, dout;
    if (dt.alloc(M * 8) || dy.alloc(M * 24) || dh.alloc(M * 8) || dout.alloc(M * 88)) return fail("device allocation failed");
    HIPCHK(hipMemcpy(dt.p, t, M * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dy.p, y, M * 24, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dh.p, h, M * 8, hipMemcpyHostToDevice));
    const dim3 grid((unsigned)((M + 63) / 64)), block(64);
    const int zm = env->d.z_simple ? ((env->d.dz == 1.0) ? 4 : 1) : 0;
    if (zm == 4)
        hipLaunchKernelGGL((pgr_step_kernel<4>), grid, block, 0, 0, env->d_dev, (const double*)dt.p, (const double*)dy.p,
                           (const double*)dh.p, M, rtol, atol, (double*)dout.p);
    else if (zm == 1)
        hipLaunchKernelGGL((pgr_step_kernel<1>), grid, block, 0, 0, env->d_dev, (const double*)dt.p, (const double*)dy.p,
                           (const double*)dh.p, M, rtol, atol, (double*)dout.p);
    else
        hipLaunchKernelGGL((pgr_step_kernel<0>), grid, block, 0, 0, env->d_dev, (const double*)dt.p, (const double*)dy.p,
                           (const double*)dh.p, M, rtol, atol, (double*)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out11, dout.p, M * 88, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int pgr_eval_points(pgr_env* env, const double* x, const double* y, int64_t M, double* out10)
{
    if (!env || !x || !y || !out10) return fail("pgr_eval_points: null argument");
    if (M <= 0) return 0;
    HIPCHK(hipSetDevice(env->device));
    DevBuf dx, dy, dout;
    if (dx.alloc(M * 8) || dy.alloc(M * 24) || dout.alloc(M * 80)) return fail("device allocation failed");
    HIPCHK(hipMemcpy(dx.p, x, M * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dy.p, y, M * 24, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pgr_eval_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, 0, env->d,
                       (const double*)dx.p, (const double*)dy.p, M, (double*)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out10, dout.p, M * 80, hipMemcpyDeviceToHost));
    return 0;
}

#endif  // PGR_DEBUG_ENTRY_H
