// pgr_hip.hip -- MI355X (gfx950) ray-fan integrator behind the C ABI of include/pgr.h.
//
// One ray per lane (wave64).  Every lane runs SciPy's adaptive Dormand-Prince 5(4)
// controller on the ray equations y = [T, z, p] over range x with the reference's
// bilinear c / dc/dz tables, +-1 step-function events located by brentq-style
// bisection on the quartic dense output, reflection at surface / bottom and the
// reference's nearest-index re-sampling -- restating, on the device,
//   REF/integration_processes.py:26-334   (derivsrd, bilinear/linear interp, events)
//   REF/launch_rays.py:325-484, 593-681    (_shoot_ray_array, _shoot_ray_segment)
//   REF/launch_rays.py:745-784             (_interpolate_ray)
//   SCIPY/rk.py:14-180,377-404,552-574 ; SCIPY/common.py:63-134 ; SCIPY/ivp.py:28-156,654-726
// (REF = /root/reference/src/pygenray, SCIPY = scipy/integrate/_ivp of SciPy 1.15.3).
//
// Layout: the c and dc/dz tables are interleaved node-wise as double2 {c, cp} so one
// 16-byte access fetches both values of a node; for range-independent tables the single
// depth profile (nz x 16 B, 96 KB at nz = 6000) is staged into LDS once per workgroup, next
// to zin and its bucket table when the depth grid is not uniform, and the bathymetry.
// State (x, y, f, h, K1..K7) lives in VGPRs.  No MFMA: there is no contraction here.
//
// Arithmetic that feeds back into the integration is written in the reference's operation
// order and compiled with -ffp-contract=off; divide / sqrt are correctly rounded, the three libm
// calls of the reference (err ** -0.2, arcsin, sin) are evaluated CORRECTLY ROUNDED
// (pgr_crmath.h) and the event locator returns brentq's own root, so the result is bit-identical
// to the CPU oracle in its correctly-rounded-libm mode (oracle/ray_oracle.c, ORC_MATH_CR).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include <float.h>
#include <string>
#include <vector>
#include <mutex>
#include <thread>
#include <atomic>
#include <initializer_list>
#include <limits>

#include "../../include/pgr.h"
#include "pgr_crmath.h"
#include "pgr_device.h"      // descriptor, arithmetic, look-ups, events, dense output, stage macros
#include "pgr_fan_kernel.h"  // the fan kernel
// the rest of this translation unit, in dependency order (one TU: the instruction-layout pass of the build works on ONE
// device object, and the kernels' order in it is part of what device_code_sha256 names)
#include "pgr_aux_kernels.h"    // wave cost / placement kernels, unit-level test kernels
#include "pgr_host.h"           // error handling, struct pgr_env, pgr_build_info
#include "pgr_env.h"            // environment construction + host-side verification, options, queries
#include "pgr_launch.h"         // wave scheduling, pgr_shoot_fan_device
#include "pgr_transfer.h"       // compaction, pipelined D2H, pgr_shoot_fan
#include "pgr_fan_handle.h"     // pgr_initial_states_device, pgr_fan_*
#include "pgr_eigen_hist.h"     // pgr_eigen_refine*, pgr_arrival_histogram_device
#include "pgr_debug_entry.h"    // pgr_debug_math / pgr_debug_step / pgr_eval_points
