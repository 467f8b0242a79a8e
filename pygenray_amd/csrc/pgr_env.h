// pgr_env.h -- the environment: tuning options, host-side construction and VERIFICATION of everything the device's look-ups rely on
// (bit-exact uniformity, not-a-knot cubic, bucket table, quadratic / cubic index estimates), upload, release, queries.
// (Part of the ONE translation unit pgr_hip.hip, included there in this order; not a stand-alone header.)
#ifndef PGR_ENV_H
#define PGR_ENV_H

extern "C" int pgr_env_set_option(pgr_env* env, int what, int a, int b)
{
    if (!env) return fail("pgr_env_set_option: null env");
    switch (what) {
    case PGR_OPT_WAVES_PER_BLOCK:
        if (a < 0 || a > 8) return fail("waves per block must be in [0,8]");
        env->waves_per_block = a;
        return 0;
    case PGR_OPT_DEPTH_SEARCH:
        if (a < 0 || a > 3) return fail("depth search: 0 = automatic, 1 = binary search, 2 = bucket table, 3 = quadratic estimate + three nodes (no cubic)");
        env->depth_search = a;
        return 0;
    case PGR_OPT_PARK:
        if (a < 1 || a > 64 || b < 0 || b > 100000) return fail("park: lanes in [1,64], trips >= 0");
        env->park_lanes = a;
        env->park_trips = b;
        return 0;
    case PGR_OPT_PLACEMENT:
        if (a < 0 || a > 2) return fail("placement: 0 = off, 1 = priorities only, 2 = placement + priorities");
        env->place = a;
        return 0;
    case PGR_OPT_API_BLOCKED:
        if (a < 0 || a > 1) return fail("api_blocked: 0 = row layout, 1 = sample-blocked kernel for HBM-table trajectory fans of pgr_shoot_fan / pgr_fan_launch");
        env->api_blocked = a;
        return 0;
    case PGR_OPT_PERSISTENT:
        if (a < 0 || a > 3) return fail("persistent: 0 = static deal of whole workgroups, 1 = persistent waves + packet queue (fans of up to two rounds: waves 4 .. 7 start at the list's cheap end), 2 = every packet from the list's head, 3 = waves 4 .. 7 always start at the cheap end");
        env->persistent = a;
        return 0;
    default:
        return fail("pgr_env_set_option: unknown option");
    }
}

// grid[j] == g0 + j*dg for all j, evaluated exactly as the device does (mul, then add)
static bool exactly_uniform(const double* g, int64_t n, double& g0, double& dg)
{
    if (n < 2) return false;
    g0 = g[0];
    dg = g[1] - g[0];
    if (!(dg > 0) || !std::isfinite(dg)) return false;
    for (int64_t j = 0; j < n; j++) {
        volatile double m = (double)j * dg;
        volatile double v = g0 + m;
        if (v != g[j]) return false;
    }
    return true;
}

// Not-a-knot cubic through (x, y): scipy.interpolate.interp1d(kind="cubic") ==
// make_interp_spline(k=3, bc_type=None) (REF/launch_rays.py:397-399).  Built in
// piecewise-polynomial form with the standard not-a-knot end rows; pp[4i..] = {y_i, s_i, c2, c3}.
static bool build_notaknot(const double* x, const double* y, int64_t n, std::vector<double>& pp)
{
    if (n < 4) return false;
    std::vector<double> dx(n), sl(n), lo(n), di(n), up(n), b(n);
    for (int64_t i = 0; i < n - 1; i++) {
        dx[i] = x[i + 1] - x[i];
        sl[i] = (y[i + 1] - y[i]) / dx[i];
    }
    for (int64_t i = 1; i < n - 1; i++) {
        lo[i] = dx[i];
        di[i] = 2 * (dx[i - 1] + dx[i]);
        up[i] = dx[i - 1];
        b[i] = 3 * (dx[i] * sl[i - 1] + dx[i - 1] * sl[i]);
    }
    double d = x[2] - x[0];
    di[0] = dx[1]; up[0] = d; lo[0] = 0;
    b[0] = ((dx[0] + 2 * d) * dx[1] * sl[0] + dx[0] * dx[0] * sl[1]) / d;
    d = x[n - 1] - x[n - 3];
    di[n - 1] = dx[n - 3]; lo[n - 1] = d; up[n - 1] = 0;
    b[n - 1] = (dx[n - 2] * dx[n - 2] * sl[n - 3] + (2 * d + dx[n - 2]) * dx[n - 3] * sl[n - 2]) / d;
    for (int64_t i = 1; i < n; i++) {
        double m = lo[i] / di[i - 1];
        di[i] -= m * up[i - 1];
        b[i] -= m * b[i - 1];
    }
    b[n - 1] /= di[n - 1];
    for (int64_t i = n - 2; i >= 0; i--) b[i] = (b[i] - up[i] * b[i + 1]) / di[i];
    pp.assign(4 * (size_t)(n - 1), 0.0);
    for (int64_t i = 0; i < n - 1; i++) {
        pp[4 * i + 0] = y[i];
        pp[4 * i + 1] = b[i];
        pp[4 * i + 2] = (3 * sl[i] - 2 * b[i] - b[i + 1]) / dx[i];
        pp[4 * i + 3] = (b[i] + b[i + 1] - 2 * sl[i]) / (dx[i] * dx[i]);
    }
    return true;
}


// Least-squares polynomial of degree `deg` (<= 3) through (x_k, y_k), x normalised to [0, 1] by the caller:
// normal equations in long double, Gaussian elimination with partial pivoting.  c[0..deg]; false if singular.
static bool polyfit_ld(const std::vector<long double>& x, const std::vector<long double>& y, int deg, long double* c)
{
    const int m = deg + 1;
    long double A[4][5] = {};
    for (size_t k = 0; k < x.size(); k++) {
        long double p[7];
        p[0] = 1;
        for (int q = 1; q <= 2 * deg; q++) p[q] = p[q - 1] * x[k];
        for (int r = 0; r < m; r++) {
            for (int q = 0; q < m; q++) A[r][q] += p[r + q];
            A[r][m] += p[r] * y[k];
        }
    }
    for (int col = 0; col < m; col++) {
        int piv = col;
        for (int r = col + 1; r < m; r++) if (fabsl(A[r][col]) > fabsl(A[piv][col])) piv = r;
        if (A[piv][col] == 0) return false;
        for (int q = 0; q <= m; q++) { long double t = A[col][q]; A[col][q] = A[piv][q]; A[piv][q] = t; }
        for (int r = 0; r < m; r++) {
            if (r == col) continue;
            const long double f = A[r][col] / A[col][col];
            for (int q = col; q <= m; q++) A[r][q] -= f * A[col][q];
        }
    }
    for (int r = 0; r < m; r++) c[r] = A[r][m] / A[r][r];
    return true;
}

// EnvDev::z_cubic: a cubic in z that estimates the node index of a smooth non-uniform depth grid to a small
// fraction of a cell, and a quadratic in the cell index for the reciprocal of the cell width.  Everything the
// device relies on is VERIFIED here, with the device's own operations (fma Horner forms), for every node / cell;
// a grid that fails any check keeps the three-node search (z_quad / z_bucket) or the binary search.
static void fit_cubic_index(const double* zin, int64_t nz, EnvDev& d)
{
    d.z_cubic = 0;
    d.zc_g0 = d.zc_g1 = d.zc_g2 = d.zc_g3 = d.zc_s0 = d.zc_s1 = d.zc_s2 = 0.0;
    if (d.z_uniform || nz < 8 || !(zin[nz - 1] > zin[0])) return;
    const long double z0 = zin[0], span = (long double)zin[nz - 1] - z0;
    std::vector<long double> u((size_t)nz), jj((size_t)nz);
    for (int64_t j = 0; j < nz; j++) { u[(size_t)j] = ((long double)zin[j] - z0) / span; jj[(size_t)j] = (long double)j; }
    long double c[4];
    if (!polyfit_ld(u, jj, 3, c)) return;
    // t(z) = sum_k c_k ((z - z0) / span)^k expanded in powers of z
    const long double a = 1 / span, b = -z0 / span;   // u = a z + b
    long double g[4];
    g[0] = c[0] + b * (c[1] + b * (c[2] + b * c[3]));
    g[1] = a * (c[1] + b * (2 * c[2] + 3 * b * c[3]));
    g[2] = a * a * (c[2] + 3 * b * c[3]);
    g[3] = a * a * a * c[3];
    double G[4] = {(double)g[0], (double)g[1], (double)g[2], (double)g[3]};
    auto idx = [&](double z) { return std::fma(z, std::fma(z, std::fma(z, G[3], G[2]), G[1]), G[0]); };
    // the estimate at the nodes: bias it down by its worst error (plus a margin that dwarfs the rounding of the
    // Horner form, ~1e-12 cells) so that t(zin[j]) <= j; with t increasing, a z of cell j then has
    // j - 1 <= t(z) < j + 1
    double worst = 0;
    for (int64_t j = 0; j < nz; j++) worst = std::fmax(worst, std::fabs(idx(zin[j]) - (double)j));
    if (!(worst <= 0.01)) return;
    const double bias = 2 * worst + 1e-7;
    G[0] -= bias;
    for (int64_t j = 0; j < nz; j++) {
        const double t = idx(zin[j]);
        if (!(t <= (double)j - 0.5e-7) || !(t >= (double)j - 0.05)) return;
        // t'(z) > 0 at every node and at the vertex of t' (a parabola: its extremum) when that lies inside the grid
        const double dt = G[1] + zin[j] * (2 * G[2] + 3 * zin[j] * G[3]);
        if (!(dt > 0)) return;
    }
    if (G[3] != 0) {
        const double zv = -G[2] / (3 * G[3]);
        if (zv > zin[0] && zv < zin[nz - 1] && !(G[1] + zv * (2 * G[2] + 3 * zv * G[3]) > 0)) return;
    }
    // reciprocal cell width as a quadratic in the cell index
    std::vector<long double> ju((size_t)nz - 1), inv((size_t)nz - 1);
    const long double jn = (long double)(nz - 2 > 0 ? nz - 2 : 1);
    for (int64_t j = 0; j + 1 < nz; j++) {
        const double den = zin[j + 1] - zin[j];
        if (!(den > 0)) return;
        ju[(size_t)j] = (long double)j / jn;
        inv[(size_t)j] = 1 / (long double)den;
    }
    long double sc[3];
    if (!polyfit_ld(ju, inv, 2, sc)) return;
    const double S[3] = {(double)sc[0], (double)(sc[1] / jn), (double)(sc[2] / (jn * jn))};
    for (int64_t j = 0; j + 1 < nz; j++) {
        const double den = zin[j + 1] - zin[j], jf = (double)j;
        const double y = std::fma(jf, std::fma(jf, S[2], S[1]), S[0]);
        if (!(std::fabs(std::fma(-den, y, 1.0)) <= 1e-8)) return;
    }
    d.z_cubic = 1;
    d.zc_g0 = G[0]; d.zc_g1 = G[1]; d.zc_g2 = G[2]; d.zc_g3 = G[3];
    d.zc_s0 = S[0]; d.zc_s1 = S[1]; d.zc_s2 = S[2];
}

template <class T>
static int upload(pgr_env* e, const T* host, size_t count, const T** dev)
{
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, count * sizeof(T)));
    e->allocs.push_back(p);
    HIPCHK(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *dev = (const T*)p;
    return 0;
}

static void env_release(pgr_env* env)
{
    (void)hipSetDevice(env->device);
    for (void* p : env->allocs) (void)hipFree(p);
    for (auto& ps : env->place_slots) {
        if (ps.ev) { (void)hipEventSynchronize(ps.ev); (void)hipEventDestroy(ps.ev); }
        if (ps.buf) (void)hipFree(ps.buf);
    }
    if (env->ws) (void)hipFree(env->ws);
    if (env->ws2) (void)hipFree(env->ws2);
    if (env->stage) (void)hipHostFree(env->stage);
    for (auto& pb : env->fan_pool) (void)hipFree(pb.first);
    if (env->stream) (void)hipStreamDestroy(env->stream);
    delete env;
}

extern "C" void pgr_env_destroy(pgr_env* env)
{
    if (!env) return;
    {
        std::lock_guard<std::mutex> lock(env->fan_pool_mutex);
        if (env->live_fans > 0) { env->doomed = true; return; }   // its fans still use its stream and tables: the last one releases it
    }
    env_release(env);
}

extern "C" int pgr_env_create(pgr_env** out, int device, const double* cin, const double* cpin,
                              const double* rin, const double* zin, int64_t nr, int64_t nz,
                              const double* depths, const double* depth_ranges,
                              const double* bottom_angles, int64_t nb)
{
    if (!out || !cin || !cpin || !rin || !zin || !depths || !depth_ranges || !bottom_angles)
        return fail("pgr_env_create: null argument");
    if (nr < 2 || nz < 2) return fail("sound speed table needs at least 2 range and 2 depth points");
    if (nr > (1 << 30) || nz > (1 << 30) || nb > (1 << 30)) return fail("table too large");
    if (nb < 4) return fail("x and y arrays must have at least 4 entries");  // interp1d(kind='cubic')
    // REF/launch_rays.py:79-90
    for (int64_t i = 1; i < nr; i++)
        if (!(rin[i] - rin[i - 1] >= 0))
            return fail("Sound speed range coordinates must be monotonically increasing.");
    for (int64_t i = 1; i < nz; i++)
        if (!(zin[i] - zin[i - 1] >= 0))
            return fail("Sound speed depth coordinates must be monotonically increasing.");
    for (int64_t i = 1; i < nb; i++)
        if (!(depth_ranges[i] - depth_ranges[i - 1] >= 0))
            return fail("Bathymetry range coordinates must be monotonically increasing.");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("pgr_env_create: no such HIP device");
    HIPCHK(hipSetDevice(device));

    pgr_env* e = new pgr_env();
    e->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        e->num_cus = prop.multiProcessorCount;
        e->max_lds = prop.sharedMemPerBlock;
        if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) e->max_lds = 160 * 1024;  // CDNA4 LDS per CU
    }
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess &&
        (size_t)optin > e->max_lds)
        e->max_lds = (size_t)optin;

    // range independence: every row bitwise equal to row 0 (both tables)
    bool indep = true;
    for (int64_t i = 1; i < nr && indep; i++)
        indep = memcmp(cin + i * nz, cin, sizeof(double) * nz) == 0 &&
                memcmp(cpin + i * nz, cpin, sizeof(double) * nz) == 0;
    e->range_indep = indep;
    size_t rows = indep ? 1 : (size_t)nr;
    std::vector<double2> tab(rows * (size_t)nz);
    for (size_t i = 0; i < rows; i++)
        for (int64_t j = 0; j < nz; j++) tab[i * nz + j] = make_double2(cin[i * nz + j], cpin[i * nz + j]);
    e->lds_path = indep && ((size_t)nz * sizeof(double2) <= e->max_lds);
    std::vector<double> pp;
    if (!build_notaknot(depth_ranges, bottom_angles, nb, pp)) {
        delete e;
        return fail("x and y arrays must have at least 4 entries");
    }
    EnvDev& d = e->d;
    int rc = 0;
#ifdef PGR_CELL_RECORDS
    if (!indep) {   // (experiment: {c, cp} of (i, j), (i, j + 1), (i + 1, j), (i + 1, j + 1) per cell, see Ctx::fetch_nodes)
        std::vector<double2> rec((size_t)(nr - 1) * (size_t)(nz - 1) * 4);
        for (int64_t i = 0; i + 1 < nr; i++)
            for (int64_t j = 0; j + 1 < nz; j++) {
                double2* q = &rec[((size_t)i * (size_t)(nz - 1) + (size_t)j) * 4];
                q[0] = tab[i * nz + j]; q[1] = tab[i * nz + j + 1]; q[2] = tab[(i + 1) * nz + j]; q[3] = tab[(i + 1) * nz + j + 1];
            }
        tab.swap(rec);
    }
#elif defined(PGR_ROW_PAIRS)
    if (!indep) {   // (experiment: {c, cp} of (i, j) and (i + 1, j) side by side, see Ctx::fetch_nodes)
        std::vector<double2> rec((size_t)(nr - 1) * (size_t)nz * 2);
        for (int64_t i = 0; i + 1 < nr; i++)
            for (int64_t j = 0; j < nz; j++) {
                rec[((size_t)i * (size_t)nz + (size_t)j) * 2] = tab[i * nz + j];
                rec[((size_t)i * (size_t)nz + (size_t)j) * 2 + 1] = tab[(i + 1) * nz + j];
            }
        tab.swap(rec);
    }
#endif
    rc |= upload(e, tab.data(), tab.size(), &d.tab);
    rc |= upload(e, rin, (size_t)nr, &d.rin);
    rc |= upload(e, zin, (size_t)nz, &d.zin);
    rc |= upload(e, depths, (size_t)nb, &d.depths);
    rc |= upload(e, depth_ranges, (size_t)nb, &d.depth_ranges);
    rc |= upload(e, pp.data(), pp.size(), &d.pp);
    if (rc) { pgr_env_destroy(e); return -1; }
    d.nr = (int)nr; d.nz = (int)nz; d.nb = (int)nb;
    d.row_stride = indep ? 0 : (int)nz;
    d.z_uniform = exactly_uniform(zin, nz, d.z0, d.dz);
    d.inv_dz = d.z_uniform ? 1.0 / d.dz : 0.0;
    d.z_pow2 = 0;
    if (d.z_uniform) {
        int ex = 0;
        bool pow2 = (std::frexp(d.dz, &ex) == 0.5);
        for (int64_t j = 0; j + 1 < nz && pow2; j++) pow2 = (zin[j + 1] - zin[j] == d.dz);
        d.z_pow2 = pow2 ? 1 : 0;
    }
    d.z_simple = (d.z_pow2 && d.z0 == 0.0) ? 1 : 0;
    d.b_zmin = depths[0];
    for (int64_t i = 1; i < nb; i++) d.b_zmin = depths[i] < d.b_zmin ? depths[i] : d.b_zmin;
    d.b_zmin -= 1.0;
    d.b_xlo = depth_ranges[0];
    d.b_xhi = depth_ranges[nb - 1];
    d.r_uniform = exactly_uniform(rin, nr, d.r0, d.dr);
    d.inv_dr = d.r_uniform ? 1.0 / d.dr : 0.0;
    d.b_uniform = exactly_uniform(depth_ranges, nb, d.b0, d.db);
    d.beta_zero = 1;
    for (double v : pp) if (v != 0.0) d.beta_zero = 0;
    d.inv_db = d.b_uniform ? 1.0 / d.db : 0.0;
    d.c_lo = cin[0]; d.c_hi = cin[0];
    for (int64_t k = 0; k < nr * nz; k++) {
        d.c_lo = cin[k] < d.c_lo ? cin[k] : d.c_lo;
        d.c_hi = cin[k] > d.c_hi ? cin[k] : d.c_hi;
    }
    if (!(d.c_lo > 0) || !std::isfinite(d.c_hi)) { d.c_lo = 0.0; d.c_hi = INFINITY; }  // no shortcut for such a table
    d.c_hi *= 1.001;
    const double tol = 1e-6;
    d.zhi_tol = zin[nz - 1] + tol;
    d.zlo_tol = zin[0] - tol;
    d.rlo_tol = rin[0] - tol;
    d.rhi_tol = rin[nr - 1] + tol;
    // bucketed depth search for a non-uniform zin (see EnvDev): bins of 0.9 min(diff(zin))
    d.z_bucket = 0; d.zbucket = nullptr; d.zb_B = 0; d.zb_z0 = 0.0; d.zb_inv_w = 0.0;
    if (!d.z_uniform && nz >= 3 && nz <= 65535) {
        double min_dz = zin[1] - zin[0];
        for (int64_t j = 1; j + 1 < nz; j++) min_dz = (zin[j + 1] - zin[j] < min_dz) ? zin[j + 1] - zin[j] : min_dz;
        const double span = zin[nz - 1] - zin[0];
        const double w = 0.9 * min_dz;
        if (min_dz > 0 && span > 0 && std::floor(span / w) + 2 <= 32768.0) {
            const int B = (int)(std::floor(span / w) + 2);
            std::vector<unsigned short> bk((size_t)B);
            bool ok = true;
            int64_t j = 0;
            for (int k = 0; k < B && ok; k++) {
                // every z the device maps to bin k (floor((z - z0) * (1/w)), two roundings) lies in [L, U)
                const double L = zin[0] + w * ((double)k - (double)(k + 1) * 1e-12);
                const double U = zin[0] + w * ((double)(k + 1) + (double)(k + 1) * 1e-12);
                while (j + 1 <= nz - 2 && zin[j + 1] < L) j++;  // j = max{ j : zin[j] < L } in [0, nz-2]
                bk[(size_t)k] = (unsigned short)j;
                if (j + 2 <= nz - 1 && !(U <= zin[j + 2])) ok = false;  // the cell is j or j+1, never beyond
            }
            if (ok && upload(e, bk.data(), bk.size(), &d.zbucket) == 0) {
                d.z_bucket = 1; d.zb_B = B; d.zb_z0 = zin[0]; d.zb_inv_w = 1.0 / w;
            }
        }
    }
    // quadratic index estimate of a smooth non-uniform zin (least squares on (u_j, j), u in [0, 1])
    d.z_quad = 0; d.zq_c0 = d.zq_c1 = d.zq_c2 = d.zq_inv_span = 0.0;
    if (!d.z_uniform && nz >= 4 && zin[nz - 1] > zin[0]) {
        const double span = zin[nz - 1] - zin[0], inv_span = 1.0 / span;
        long double S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, T0 = 0, T1 = 0, T2 = 0;
        for (int64_t j = 0; j < nz; j++) {
            const long double u = (long double)((zin[j] - zin[0]) * inv_span), y = (long double)j;
            S0 += 1; S1 += u; S2 += u * u; S3 += u * u * u; S4 += u * u * u * u;
            T0 += y; T1 += y * u; T2 += y * u * u;
        }
        // normal equations [S0 S1 S2; S1 S2 S3; S2 S3 S4] c = [T0 T1 T2] by Cramer's rule
        const long double D = S0 * (S2 * S4 - S3 * S3) - S1 * (S1 * S4 - S3 * S2) + S2 * (S1 * S3 - S2 * S2);
        if (D != 0) {
            const double c0 = (double)((T0 * (S2 * S4 - S3 * S3) - S1 * (T1 * S4 - S3 * T2) + S2 * (T1 * S3 - S2 * T2)) / D);
            const double c1 = (double)((S0 * (T1 * S4 - T2 * S3) - T0 * (S1 * S4 - S3 * S2) + S2 * (S1 * T2 - S2 * T1)) / D);
            const double c2 = (double)((S0 * (S2 * T2 - S3 * T1) - S1 * (S1 * T2 - S2 * T1) + T0 * (S1 * S3 - S2 * S2)) / D);
            bool ok = (c1 > 0) && (c1 + 2 * c2 > 0);   // g' > 0 on [0, 1]
            for (int64_t j = 0; j < nz && ok; j++) {
                volatile double u = (zin[j] - zin[0]) * inv_span;   // the device's own arithmetic
                volatile double q = c1 + u * c2;
                volatile double g = c0 + u * q;
                ok = std::fabs((double)g - (double)j) <= 0.45;
            }
            if (ok) { d.z_quad = 1; d.zq_c0 = c0; d.zq_c1 = c1; d.zq_c2 = c2; d.zq_inv_span = inv_span; d.zb_z0 = zin[0]; }
        }
    }
    fit_cubic_index(zin, nz, d);
    if (upload(e, &e->d, 1, &e->d_dev)) { pgr_env_destroy(e); return -1; }
    *out = e;
    return 0;
}

static bool blocked_layout_fits(const pgr_env* env);   // (pgr_launch.h)

extern "C" int pgr_env_query(const pgr_env* env, int what)
{
    if (!env) return fail("null env");
    switch (what) {
    case 0: return env->range_indep;
    case 1: return env->d.z_uniform;
    case 2: return env->d.r_uniform;
    case 3: return env->lds_path;
    case 4: return env->device;
    case 5: return env->d.z_cubic;
    case 6: return env->d.z_quad;
    case 7: return env->d.z_bucket;
    case 8: return blocked_layout_fits(env) ? 1 : 0;   // a trajectory fan of this environment would take the sample-blocked kernel (tables in HBM / L2, LDS left for the staging)
    default: return fail("pgr_env_query: unknown property");
    }
}

#endif  // PGR_ENV_H
