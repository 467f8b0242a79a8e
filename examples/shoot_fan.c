/* Plain-C use of the C ABI (include/pgr.h): a 64-ray Munk fan to 100 km, end states only, through the host-pointer
 * entry; then the same fan device resident (pgr_fan_*), its depth trajectories fetched afterwards.
 *
 *   gcc -std=c99 -Iinclude examples/shoot_fan.c -o shoot_fan -Lpygenray_amd/csrc -lpgr_hip -lm \
 *       -Wl,-rpath,$PWD/pygenray_amd/csrc
 *
 * The environment is the 7-array contract pygenray ships to its pool workers
 * (REF/multi_processing.py:37-45); y0 = [0, z_s, sin(theta)/c(x_s, z_s)] (REF/launch_rays.py:140-144). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "pgr.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static double munk(double z)
{
    double zh = 2.0 * (z - 1300.0) / 1300.0;
    return 1500.0 * (1.0 + 0.00737 * (zh - 1.0 + exp(-zh)));
}

int main(void)
{
    enum { NR = 20, NZ = 6000, NB = 20, N = 64 };
    double *cin = malloc(sizeof(double) * NR * NZ), *cpin = malloc(sizeof(double) * NR * NZ);
    double rin[NR], zin[NZ], depths[NB], depth_ranges[NB], bottom_angles[NB];
    for (int j = 0; j < NZ; j++) zin[j] = (double)j;
    for (int i = 0; i < NR; i++) {
        rin[i] = 100e3 * i / (NR - 1);
        depth_ranges[i] = rin[i];
        depths[i] = 5000.0;
        bottom_angles[i] = 0.0;
        for (int j = 0; j < NZ; j++) cin[i * NZ + j] = munk(zin[j]);
        for (int j = 0; j < NZ; j++) {  /* np.gradient, edge_order 1 */
            int a = j > 0 ? j - 1 : 0, b = j < NZ - 1 ? j + 1 : NZ - 1;
            cpin[i * NZ + j] = (cin[i * NZ + b] - cin[i * NZ + a]) / (zin[b] - zin[a]);
        }
    }
    pgr_env* env = NULL;
    if (pgr_env_create(&env, 0, cin, cpin, rin, zin, NR, NZ, depths, depth_ranges, bottom_angles, NB)) {
        fprintf(stderr, "pgr_env_create: %s\n", pgr_last_error());
        return 1;
    }
    double y0[N][3], end[N][3];
    int32_t nb[N], ns[N], st[N], nsteps[N];
    const double zs = 1000.0, c0 = munk(zs);
    for (int k = 0; k < N; k++) {
        double th = (-15.0 + 30.0 * k / (N - 1)) * M_PI / 180.0;
        y0[k][0] = 0.0; y0[k][1] = zs; y0[k][2] = sin(th) / c0;
    }
    int rc = pgr_shoot_fan(env, &y0[0][0], N, 0.0, 100e3, NULL, 0, 1e-9, 1e-6, PGR_TERMINATE_BACKWARDS,
                           1000000, NULL, NULL, NULL, &end[0][0], nb, ns, st, nsteps, NULL);
    if (rc) {
        fprintf(stderr, "pgr_shoot_fan: %s\n", pgr_last_error());
        return 1;
    }
    for (int k = 0; k < N; k += 9)
        printf("ray %2d: status %d, T = %.9f s, z = %.6f m, %d steps, %d bottom / %d surface bounces\n", k, st[k],
               end[k][0], end[k][1], nsteps[k], nb[k], ns[k]);
    /* The same fan with its results left in HBM (pgr_fan_*): the call returns while the kernel runs; the depth
     * trajectories of the surviving rays are fetched afterwards, on their own ([S][M], PGR_COMPACT). */
    enum { S = 11 };
    pgr_fan* fan = NULL;
    if (pgr_fan_launch(env, &y0[0][0], NULL, 0.0, 0.0, N, 0.0, 100e3, S, 1e-9, 1e-6, PGR_TERMINATE_BACKWARDS, 1000000, &fan)) {
        fprintf(stderr, "pgr_fan_launch: %s\n", pgr_last_error());
        return 1;
    }
    int64_t n_rays = 0, n_ok = 0;
    double* zt = malloc(sizeof(double) * S * N);
    double end2[N][3];
    if (pgr_fan_wait(fan, &n_rays, &n_ok) || pgr_fan_fetch_rays(fan, &end2[0][0], NULL, NULL, NULL, NULL, NULL) ||
        pgr_fan_fetch_samples(fan, NULL, zt, NULL, PGR_COMPACT)) {
        fprintf(stderr, "pgr_fan_*: %s\n", pgr_last_error());
        return 1;
    }
    int same = (n_rays == N);
    for (int k = 0; k < N; k++) same = same && end2[k][0] == end[k][0] && end2[k][1] == end[k][1] && end2[k][2] == end[k][2];
    /* all rays survive here, so column k of the compacted [S][M] block is ray k: its last row is the exact end state */
    for (int k = 0; k < N && n_ok == N; k++) same = same && zt[(S - 1) * n_ok + k] == end[k][1];
    printf("device-resident fan: %lld of %lld rays kept, %d samples each, equal to the host-entry fan: %s\n", (long long)n_ok,
           (long long)n_rays, (int)S, same ? "yes" : "NO");
    free(zt);
    pgr_fan_destroy(fan);
    pgr_env_destroy(env);
    free(cin); free(cpin);
    return same ? 0 : 2;
}
