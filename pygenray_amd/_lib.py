"""ctypes binding of the C ABI in include/pgr.h (libpgr_hip.so).

The product path is HIP only: if the library is missing or cannot be loaded this module
raises -- there is no CPU fallback (the CPU oracle under oracle/ is test infrastructure and
is never imported from here).
"""
import ctypes
import os
import sys
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
REFERENCE_LIB = os.path.join(CSRC, "libpgr_hip.so")
CONTRACTED_LIB = os.path.join(CSRC, "libpgr_hip_fma.so")
CONTRACTED_FLAGS = ["-DPGR_FMA", "-ffp-contract=fast"]


class PgrError(RuntimeError):
    pass


# Which arithmetic this PROCESS computes in -- an import-time choice, one library per process:
#   PGR_ARITH=reference (default)  libpgr_hip.so: the reference's IEEE operations in the reference's order, correctly
#                                  rounded div / sqrt / pow / asin / sin -- bit-identical to the CPU oracle; every parity
#                                  statement of this package is about THIS build;
#   PGR_ARITH=contracted           libpgr_hip_fma.so: the same sources with FMA contraction allowed (a*b + c fused, 2-ulp
#                                  reciprocal square root): ~10 % faster, statistically as close to pygenray as pygenray is
#                                  to itself (its numba kernels are fastmath=True, REF/integration_processes.py:26, and
#                                  SciPy's stage sums run through BLAS), but NOT the oracle's bits: no bit-parity claim.
ARITH = os.environ.get("PGR_ARITH", "reference").strip().lower() or "reference"
if ARITH not in ("reference", "contracted"):
    raise PgrError(f"PGR_ARITH={ARITH!r}: expected 'reference' (default) or 'contracted'")
LIB_PATH = CONTRACTED_LIB if ARITH == "contracted" else REFERENCE_LIB
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -disable-machine-licm: the bounce code's asin / sin / pow are inlined polynomials; machine LICM
# hoists their ~60 fp64 coefficients out of the WHOLE step loop into registers (235-256 VGPRs, 140-170
# SGPR spills); without it the kernel needs 147-162 VGPRs and 33-92 SGPR spills, and the step loop,
# whose own constants never fitted anyway, gets up to 3 % faster.  (Three waves per SIMD then fit,
# but measured slower: 1e6 rays 45.0 vs 42.4 ms, 180 000 rays 13.1 vs 9.8 ms -- a workgroup holds
# its CU until its last wave ends and the fans are VALU-bound already.)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-disable-machine-licm"]
if os.environ.get("PGR_FMA"):      # experiments only: FMA contraction + 2-ulp rsqrt (breaks 1e-8 parity)
    HIPCC_FLAGS += ["-DPGR_FMA", "-ffp-contract=fast"]
elif os.environ.get("PGR_STRICT"):  # compiler's IEEE divide/sqrt and pow()
    HIPCC_FLAGS += ["-DPGR_STRICT", "-ffp-contract=off"]
else:                               # default: reference arithmetic, cheaper correctly-rounded div/sqrt
    HIPCC_FLAGS += ["-ffp-contract=off"]

PGR_TERMINATE_BACKWARDS = 1
PGR_SAMPLE_MAJOR = 2
PGR_EXACT_BISECTION = 4
PGR_EXACT_SAMPLES = 32
PGR_STORED_SIGN = 64
PGR_COMPACT = 128
PGR_PACKED_END = 256
PGR_SAVE_LINSPACE = 8
PGR_SKIP_NAN_Y0 = 512
PGR_LAUNCH_SLOWNESS = 1024
PGR_SAMPLE_BLOCKED = 2048

RAY_STATUS = {0: "ok", 1: "vertical", 2: "bbox", 3: "backward", 4: "step_too_small",
              5: "max_steps", 6: "bottom_angle_range", 7: "event_error", 8: "skipped"}

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_i64 = ctypes.c_int64
_vp = ctypes.c_void_p

_lib = None


def _llvm_bin():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "lib", "llvm", "bin")


def _relayout(td, verbose):
    """Second pass of the build (pygenray_amd/_isa_layout.py): re-encode 4-byte VALU instructions as
    8-byte ones where that keeps 8-byte instructions from straddling a 32-byte fetch window, then
    assemble, link and bundle the device code again and put it into the host object.  Works on the
    intermediates hipcc -save-temps left in `td`; returns the path of the finished library."""
    import re
    from . import _isa_layout
    B = _llvm_bin()
    dev_s = os.path.join(td, "pgr_hip-hip-amdgcn-amd-amdhsa-gfx950.s")
    dev_o = os.path.join(td, "pgr_hip-hip-amdgcn-amd-amdhsa-gfx950.o")
    host_s = os.path.join(td, "pgr_hip-host-x86_64-unknown-linux-gnu.s")
    layout = _isa_layout.object_layout(os.path.join(B, "llvm-objdump"), dev_o)
    with open(dev_s) as f:
        text, report = _isa_layout.relayout(f.read(), layout)
    dev2_s, dev2_o = os.path.join(td, "dev2.s"), os.path.join(td, "dev2.o")
    dev2_out, fb = os.path.join(td, "dev2.out"), os.path.join(td, "dev2.hipfb")
    with open(dev2_s, "w") as f:
        f.write(text)
    run = lambda c: subprocess.check_call(c, cwd=td)
    run([os.path.join(B, "clang"), "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj",
         "-main-file-name", "pgr_hip.hip", "-target-cpu", "gfx950", "-mrelocation-model", "pic",
         "-o", dev2_o, dev2_s])
    # the re-encoded object must hold the same instructions in the same order
    after = _isa_layout.object_layout(os.path.join(B, "llvm-objdump"), dev2_o)
    _isa_layout.check_same_program(layout, after)
    run([os.path.join(B, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared",
         "-plugin-opt=-amdgpu-internalize-symbols", "-plugin-opt=mcpu=gfx950", "-plugin-opt=O3",
         "--whole-archive", "-o", dev2_out, dev2_o, "--no-whole-archive"])
    run([os.path.join(B, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
         "-input=/dev/null", f"-input={dev2_out}", f"-output={fb}"])
    # the host assembly carries the fat binary as one string: point it at the new bundle instead
    with open(host_s, "rb") as f:
        h = f.read()
    i = h.index(b'\t.asciz\t"__CLANG_OFFLOAD_BUNDLE__')
    j = h.index(b"\n", i)
    k = h.index(b"\n", j + 1)
    m = re.match(rb"\t\.size\t(\S+), (\d+)", h[j + 1:k])
    if not m:
        raise RuntimeError("unexpected layout of the fat binary in the host assembly")
    h = (h[:i] + b'\t.incbin\t"' + fb.encode() + b'"\n\t.size\t' + m.group(1) + b", " +
         str(os.path.getsize(fb)).encode() + h[k:])
    # ... and say so in the library: pgr_build_info() reads this tag (same length, so nothing moves)
    b = sum(v[0] for v in report.values()); a = sum(v[1] for v in report.values())
    tag0 = b"PGR_BUILD_TAG:plain hipcc"
    k0 = h.index(tag0)
    k1 = h.index(b'"', k0)
    tag = (f"PGR_BUILD_TAG:relaid, {b} -> {a} straddling 8-byte instructions").encode()
    if len(tag) > k1 - k0:
        raise RuntimeError("build tag too long")
    h = h[:k0] + tag + b" " * (k1 - k0 - len(tag)) + h[k1:]
    host2_s, host2_o = os.path.join(td, "host2.s"), os.path.join(td, "host2.o")
    with open(host2_s, "wb") as f:
        f.write(h)
    run([os.path.join(B, "clang"), "-c", host2_s, "-o", host2_o])
    out = os.path.join(td, "relaid.so")
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, host2_o])
    if verbose:
        print(f"instruction layout: {b} -> {a} 8-byte instructions straddle a 32-byte fetch window "
              f"({sum(v[2] for v in report.values())} re-encoded as e64)")
    return out


def build(force=False, verbose=False, out=None, extra_flags=()):
    """Compile csrc/pgr_hip.hip (ONE translation unit; it includes every csrc/*.h) for gfx950 (cross-compiles without a GPU): hipcc, then the
    instruction-layout pass over its assembly (_relayout; PGR_NO_RELAYOUT=1 or any failure of that
    pass leaves the plain hipcc build in place).  `out` / `extra_flags`: build a variant library
    somewhere else (A/B experiments, scripts/kbench.py --lib); the default builds the product."""
    import shutil
    import tempfile
    if out is None and ARITH == "contracted" and not extra_flags:
        extra_flags = CONTRACTED_FLAGS      # (PGR_ARITH=contracted: `build()` builds the library this process loads)
    LIB_PATH = out or globals()["LIB_PATH"]
    src = os.path.join(CSRC, "pgr_hip.hip")
    hdr = os.path.join(_HERE, "..", "include", "pgr.h")
    # (one translation unit: pgr_hip.hip includes every csrc/*.h -- device building blocks, the fan kernel, the host side in pieces)
    deps = [src, hdr, os.path.join(_HERE, "_isa_layout.py")] + sorted(
        os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h"))
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
            return LIB_PATH
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"   # never leave a half-written library behind
    td = tempfile.mkdtemp(prefix="pgr_build_")
    try:
        plain = os.path.join(td, "plain.so")
        cmd = [HIPCC] + HIPCC_FLAGS + list(extra_flags) + ["-save-temps", "-o", plain, src]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=td)
        result = plain
        if not os.environ.get("PGR_NO_RELAYOUT"):
            try:
                result = _relayout(td, verbose)
            except Exception as e:  # the plain build is complete and correct: keep it, say why
                print(f"pygenray_amd: instruction-layout pass skipped ({type(e).__name__}: {e})", file=sys.stderr)
        shutil.copyfile(result, tmp)
        os.chmod(tmp, 0o755)
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
        shutil.rmtree(td, ignore_errors=True)
    return LIB_PATH


def build_contracted(force=False, verbose=False):
    """The PGR_ARITH=contracted library (libpgr_hip_fma.so) beside the product: same sources, FMA contraction allowed."""
    return build(force=force, verbose=verbose, out=CONTRACTED_LIB, extra_flags=CONTRACTED_FLAGS)


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two
    HIP runtimes in one process do not work (the second one finds no GPUs, and a torch stream
    handed to the other runtime is undefined behaviour), so when torch is installed its copy is
    loaded first and libpgr_hip.so's DT_NEEDED libamdhip64.so.7 resolves to it -- whichever of
    torch / pygenray_amd is imported first.  Without torch the system runtime is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin:
            p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(p):
                ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load libpgr_hip.so; raise loudly if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise PgrError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc"
            + ("; PGR_ARITH=contracted loads the FMA-contracted variant, built by the same call or by "
               "pygenray_amd._lib.build_contracted()" if ARITH == "contracted" else "") +
            "). pygenray_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    L.pgr_last_error.restype = ctypes.c_char_p
    L.pgr_device_count.restype = ctypes.c_int
    L.pgr_env_create.restype = ctypes.c_int
    L.pgr_env_create.argtypes = [ctypes.POINTER(_vp), ctypes.c_int, _dp, _dp, _dp, _dp, _i64, _i64,
                                 _dp, _dp, _dp, _i64]
    L.pgr_env_destroy.restype = None
    L.pgr_env_destroy.argtypes = [_vp]
    L.pgr_env_query.restype = ctypes.c_int
    L.pgr_env_query.argtypes = [_vp, ctypes.c_int]
    fan_common = [_vp, _vp, _i64, ctypes.c_double, ctypes.c_double, _vp, ctypes.c_int32,
                  ctypes.c_double, ctypes.c_double, ctypes.c_uint32, _i64, _vp, _vp, _vp, _vp, _vp,
                  _vp, _vp, _vp, _vp]
    L.pgr_shoot_fan.restype = ctypes.c_int
    L.pgr_shoot_fan.argtypes = fan_common
    L.pgr_shoot_fan_device.restype = ctypes.c_int
    L.pgr_shoot_fan_device.argtypes = fan_common + [_vp]
    L.pgr_env_set_option.restype = ctypes.c_int
    L.pgr_env_set_option.argtypes = [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.pgr_eval_points.restype = ctypes.c_int
    L.pgr_eval_points.argtypes = [_vp, _dp, _dp, _i64, _dp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise PgrError(load().pgr_last_error().decode())


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def _vptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class EnvHandle:
    """Owns one pgr_env (environment tables resident in HBM on `device`)."""

    def __init__(self, cin, cpin, rin, zin, depths, depth_ranges, bottom_angles, device=0):
        L = load()
        cin, cpin, rin, zin = _c(cin), _c(cpin), _c(rin), _c(zin)
        depths, depth_ranges, bottom_angles = _c(depths), _c(depth_ranges), _c(bottom_angles)
        if cin.ndim != 2 or cin.shape != cpin.shape or cin.shape != (len(rin), len(zin)):
            raise ValueError("cin/cpin must have shape (len(rin), len(zin))")
        if not (len(depths) == len(depth_ranges) == len(bottom_angles)):
            raise ValueError("depths, depth_ranges and bottom_angles must have equal length")
        h = _vp()
        check(L.pgr_env_create(ctypes.byref(h), int(device), _p(cin), _p(cpin), _p(rin), _p(zin),
                               len(rin), len(zin), _p(depths), _p(depth_ranges), _p(bottom_angles),
                               len(depths)))
        self._h = h
        self.device = int(device)
        self.shape = cin.shape

    def query(self, what):
        return load().pgr_env_query(self._h, int(what))

    # tuning options of this environment (include/pgr.h; results never depend on them)
    _OPTIONS = {"waves_per_block": 0, "depth_search": 1, "park": 2, "placement": 3, "persistent": 4, "api_blocked": 5}

    def set_option(self, name, a, b=0):
        check(load().pgr_env_set_option(self._h, self._OPTIONS[name], int(a), int(b)))

    @property
    def range_independent(self):
        return bool(self.query(0))

    @property
    def lds_path(self):
        return bool(self.query(3))

    @property
    def blocked_layout(self):
        """True when a sample-major trajectory fan of this environment is best written sample-blocked ([ceil(S/4)][N][4],
        PGR_SAMPLE_BLOCKED): its tables stay in HBM / L2 and the LDS has room for the staging (include/pgr.h)."""
        return bool(self.query(8))

    def close(self):
        if getattr(self, "_h", None):
            load().pgr_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-pointer entry (NumPy in / NumPy out) ----
    def shoot_fan(self, y0, source_range, receiver_range, num_range_save, rtol=1e-9, atol=1e-6,
                  terminate_backwards=True, max_steps=1_000_000, save=True, sample_major=False,
                  exact_bisection=False, exact_samples=False, stored_sign=False, compact=False, buffers=None):
        """``buffers``: (T, Z, P) float64 arrays of the right shape to write into (caller-owned, e.g.
        reused across fans) instead of fresh ones."""
        L = load()
        y0 = _c(y0).reshape(-1, 3)
        N, S = len(y0), int(num_range_save)
        r = np.linspace(source_range, receiver_range, S)
        flags = (PGR_TERMINATE_BACKWARDS if terminate_backwards else 0) | \
            (PGR_SAMPLE_MAJOR if sample_major else 0) | (PGR_EXACT_BISECTION if exact_bisection else 0) | \
            (PGR_EXACT_SAMPLES if exact_samples else 0) | (PGR_STORED_SIGN if stored_sign else 0) | \
            (PGR_COMPACT if (compact and sample_major) else 0)
        if save:
            shape = (S, N) if sample_major else (N, S)
            if buffers is not None:
                T, Z, P = buffers
                for b_ in (T, Z, P):
                    if b_.shape != shape or b_.dtype != np.float64 or not b_.flags.c_contiguous:
                        raise ValueError(f"buffers must be C-contiguous float64 arrays of shape {shape}")
            else:
                T = np.empty(shape); Z = np.empty(shape); P = np.empty(shape)
        else:
            T = Z = P = None
        end = np.empty((N, 3))
        nb = np.zeros(N, np.int32); ns = np.zeros(N, np.int32); st = np.zeros(N, np.int32)
        nsteps = np.zeros(N, np.int32); nrej = np.zeros(N, np.int32)
        check(L.pgr_shoot_fan(self._h, _vptr(y0), N, float(source_range), float(receiver_range),
                              _vptr(r), S, float(rtol), float(atol), flags, int(max_steps),
                              _vptr(T), _vptr(Z), _vptr(P), _vptr(end), _vptr(nb), _vptr(ns),
                              _vptr(st), _vptr(nsteps), _vptr(nrej)))
        if save and compact and sample_major:
            # PGR_COMPACT: the trajectories of the M rays with status 0 sit as [S][M] at the start
            # of the buffers
            M = int(np.count_nonzero(st == 0))
            if M < N:
                T, Z, P = (a.reshape(-1)[:S * M].reshape(S, M) for a in (T, Z, P))
        return dict(r=r, T=T, z=Z, p=P, end=end, n_bott=nb, n_surf=ns, status=st, n_steps=nsteps,
                    n_rej=nrej)

    # ---- device-pointer entry (integers are raw device addresses, e.g. tensor.data_ptr()) ----
    def shoot_fan_device(self, y0_ptr, N, source_range, receiver_range, r_ptr, S, rtol, atol, flags,
                         max_steps, T_ptr, Z_ptr, P_ptr, end_ptr, nb_ptr, ns_ptr, st_ptr,
                         nsteps_ptr, nrej_ptr, stream=0):
        L = load()
        v = lambda q: ctypes.c_void_p(q) if q else None  # noqa: E731
        check(L.pgr_shoot_fan_device(self._h, v(y0_ptr), int(N), float(source_range),
                                     float(receiver_range), v(r_ptr), int(S), float(rtol),
                                     float(atol), int(flags), int(max_steps), v(T_ptr), v(Z_ptr),
                                     v(P_ptr), v(end_ptr), v(nb_ptr), v(ns_ptr), v(st_ptr),
                                     v(nsteps_ptr), v(nrej_ptr), v(stream)))

    def debug_step(self, t, y, h, rtol=1e-9, atol=1e-6):
        """One RK45 step attempt per (t, y, h): y_new[3], f_new[3], error_norm, 0.9 err**-0.2, f[3]."""
        t = _c(t); y = _c(y).reshape(-1, 3); h = _c(h)
        out = np.empty((len(t), 11))
        L = load()
        L.pgr_debug_step.restype = ctypes.c_int
        L.pgr_debug_step.argtypes = [_vp, _dp, _dp, _dp, _i64, ctypes.c_double, ctypes.c_double, _dp]
        check(L.pgr_debug_step(self._h, _p(t), _p(y), _p(h), len(t), float(rtol), float(atol), _p(out)))
        return out

    def eigen_refine(self, th1, th2, z1, z2, receiver_depth, source_depth, source_range, receiver_range, c_source,
                     rtol=1e-9, atol=1e-6, terminate_backwards=True, max_steps=1_000_000, ztol=1.0, max_iter=20,
                     slowness=None):
        """pgr_eigen_refine_depths_fn: the false-position loop of REF/eigenrays.py:206-268 for all brackets, on the device.
        `receiver_depth`: one depth for all brackets, or one per bracket (the brackets of several receiver depths
        searched together).  `slowness(ode_angles_deg) -> p0`: the caller's sin(radians(.)) / c for the trial rays (the shim
        passes NumPy's, the reference's arithmetic); None: the device's correctly rounded sine.  The callback runs inside
        the search with this environment's workspace lock held: it must not shoot rays or search on THIS EnvHandle
        (include/pgr.h); plain NumPy arithmetic, as the shim's, is what it is for."""
        L = load()
        th1, th2, z1, z2 = (_c(a).reshape(-1) for a in (th1, th2, z1, z2))
        n = len(th1)
        rd = _c(np.broadcast_to(np.asarray(receiver_depth, dtype=float), (n,)))
        theta = np.full(n, np.nan); zend = np.full(n, np.nan); tend = np.full(n, np.nan)
        state = np.zeros(n, np.int32); ntrial = np.zeros(n, np.int32)
        launches = ctypes.c_int32(0)
        FN = ctypes.CFUNCTYPE(None, _dp, _i64, _dp, _vp)
        L.pgr_eigen_refine_depths_fn.restype = ctypes.c_int
        L.pgr_eigen_refine_depths_fn.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp, _vp] + [ctypes.c_double] * 6 + [
            ctypes.c_uint32, _i64, ctypes.c_double, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_int32),
            FN, _vp]
        failure = []

        def _cb(ang_p, m, out_p, _user):
            try:    # (an exception must not unwind through the C frames: it is re-raised when the call has returned)
                ang = np.ctypeslib.as_array(ang_p, shape=(m,))
                np.ctypeslib.as_array(out_p, shape=(m,))[:] = slowness(ang)
            except BaseException as exc:   # noqa: BLE001
                failure.append(exc)
                np.ctypeslib.as_array(out_p, shape=(m,))[:] = np.nan
        cb = FN(_cb) if slowness is not None else ctypes.cast(None, FN)
        check(L.pgr_eigen_refine_depths_fn(self._h, n, _vptr(th1), _vptr(th2), _vptr(z1), _vptr(z2), _vptr(rd),
                                           float(source_depth), float(source_range), float(receiver_range), float(c_source),
                                           float(rtol), float(atol), PGR_TERMINATE_BACKWARDS if terminate_backwards else 0,
                                           int(max_steps), float(ztol), int(max_iter), _vptr(theta), _vptr(state), _vptr(ntrial),
                                           _vptr(zend), _vptr(tend), ctypes.byref(launches), cb, None))
        if failure:
            raise failure[0]
        return dict(theta=theta, state=state, n_trial=ntrial, z_end=zend, t_end=tend, launches=int(launches.value))

    def last_instance(self):
        """pgr_debug_last_instance: dict(lds_tab, zm, save, persist, blocks, threads, lds_bytes, queue_tail) of the last fan launch."""
        L = load()
        L.pgr_debug_last_instance.restype = ctypes.c_int
        L.pgr_debug_last_instance.argtypes = [_vp, ctypes.POINTER(ctypes.c_int32)]
        out = (ctypes.c_int32 * 8)()
        check(L.pgr_debug_last_instance(self._h, out))
        return dict(zip(("lds_tab", "zm", "save", "persist", "blocks", "threads", "lds_bytes", "queue_tail"), (int(v) for v in out)))

    def eval_points(self, x, y):
        x = _c(x); y = _c(y).reshape(-1, 3)
        out = np.empty((len(x), 10))
        check(load().pgr_eval_points(self._h, _p(x), _p(y), len(x), _p(out)))
        return out


class FanHandle:
    """A fan whose results stay in HBM (pgr_fan_*, include/pgr.h): launched in the constructor (returns while the
    kernel runs), per-ray arrays and trajectories fetched on demand."""

    def __init__(self, env, x0, x1, S, y0=None, ode_angles_deg=None, source_depth=0.0, c_source=1.0, rtol=1e-9,
                 atol=1e-6, terminate_backwards=True, max_steps=1_000_000, stored_sign=False, exact_samples=False,
                 exact_bisection=False, p0=None, skip_nan=False):
        L = load()
        L.pgr_fan_launch.restype = ctypes.c_int
        L.pgr_fan_launch.argtypes = [_vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _i64, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_uint32, _i64, ctypes.POINTER(_vp)]
        L.pgr_fan_wait.restype = ctypes.c_int
        L.pgr_fan_wait.argtypes = [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]
        L.pgr_fan_fetch_rays.restype = ctypes.c_int
        L.pgr_fan_fetch_rays.argtypes = [_vp] * 7
        L.pgr_fan_fetch_samples.restype = ctypes.c_int
        L.pgr_fan_fetch_samples.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_uint32]
        L.pgr_fan_destroy.restype = None
        L.pgr_fan_destroy.argtypes = [_vp]
        self._env = env   # keeps the environment (its stream, its tables) alive
        if y0 is not None:
            y0 = _c(y0).reshape(-1, 3)
            n = len(y0)
        elif p0 is not None:      # the caller's own sin(radians(angle)) / c per ray (PGR_LAUNCH_SLOWNESS)
            ode_angles_deg = _c(p0).reshape(-1)
            n = len(ode_angles_deg)
        else:
            ode_angles_deg = _c(ode_angles_deg).reshape(-1)
            n = len(ode_angles_deg)
        self.N, self.S = n, int(S)
        flags = (PGR_TERMINATE_BACKWARDS if terminate_backwards else 0) | (PGR_STORED_SIGN if stored_sign else 0) | \
            (PGR_EXACT_SAMPLES if exact_samples else 0) | (PGR_EXACT_BISECTION if exact_bisection else 0) | \
            (PGR_LAUNCH_SLOWNESS if p0 is not None else 0) | (PGR_SKIP_NAN_Y0 if skip_nan else 0)
        h = _vp()
        check(L.pgr_fan_launch(env._h, _vptr(y0), _vptr(ode_angles_deg), float(source_depth), float(c_source), n,
                               float(x0), float(x1), self.S, float(rtol), float(atol), flags, int(max_steps),
                               ctypes.byref(h)))
        self._h = h
        self.M = None

    def wait(self):
        n, m = _i64(0), _i64(0)
        check(load().pgr_fan_wait(self._h, ctypes.byref(n), ctypes.byref(m)))
        self.M = int(m.value)
        return int(n.value), self.M

    def fetch_rays(self):
        n = self.N
        end = np.empty((n, 3))
        nb = np.empty(n, np.int32); ns = np.empty(n, np.int32); st = np.empty(n, np.int32)
        n1 = np.empty(n, np.int32); n2 = np.empty(n, np.int32)
        check(load().pgr_fan_fetch_rays(self._h, _vptr(end), _vptr(nb), _vptr(ns), _vptr(st), _vptr(n1), _vptr(n2)))
        self.M = int(np.count_nonzero(st == 0))
        return dict(end=end, n_bott=nb, n_surf=ns, status=st, n_steps=n1, n_rej=n2)

    def fetch_rays_compact(self, per_ray=None):
        """The surviving rays only (launch order): end [M, 3], n_bott / n_surf [M] int64 and, squeezed the same way,
        the caller's per-ray array `per_ray` [N] -> [M]."""
        L = load()
        L.pgr_fan_fetch_rays_compact.restype = ctypes.c_int
        L.pgr_fan_fetch_rays_compact.argtypes = [_vp] * 6
        if self.M is None:
            self.wait()
        m = self.M
        end = np.empty((m, 3)); nb = np.empty(m, np.int64); ns = np.empty(m, np.int64)
        src = None if per_ray is None else _c(per_ray).reshape(-1)
        out = None if per_ray is None else np.empty(m)
        check(L.pgr_fan_fetch_rays_compact(self._h, _vptr(src), _vptr(out), _vptr(end), _vptr(nb), _vptr(ns)))
        return dict(end=end, n_bott=nb, n_surf=ns, per_ray=out)

    def status(self):
        """status [N] (waits for the kernel)."""
        st = np.empty(self.N, np.int32)
        check(load().pgr_fan_fetch_rays(self._h, None, None, None, _vptr(st), None, None))
        self.M = int(np.count_nonzero(st == 0))
        return st

    def fetch_samples(self, which=("T", "z", "p"), compact=True):
        """-> dict name -> (S, M) array (M = surviving rays when `compact`, else all N; dropped rays are NaN then)."""
        if self.M is None:
            self.wait()
        cols = self.M if compact else self.N
        bufs = {k: (np.empty((self.S, self.N)) if k in which else None) for k in ("T", "z", "p")}
        check(load().pgr_fan_fetch_samples(self._h, _vptr(bufs["T"]), _vptr(bufs["z"]), _vptr(bufs["p"]),
                                           PGR_COMPACT if compact else 0))
        return {k: (v.reshape(-1)[:self.S * cols].reshape(self.S, cols) if cols != self.N else v)
                for k, v in bufs.items() if v is not None}

    def close(self):
        if getattr(self, "_h", None):
            load().pgr_fan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def initial_states_device(device, ang_ptr, n, source_depth, c_source, y0_ptr, stream=0):
    """pgr_initial_states_device on raw device pointers (ints)."""
    L = load()
    L.pgr_initial_states_device.restype = ctypes.c_int
    L.pgr_initial_states_device.argtypes = [ctypes.c_int, _vp, _i64, ctypes.c_double, ctypes.c_double, _vp, _vp]
    check(L.pgr_initial_states_device(int(device), _vp(ang_ptr), int(n), float(source_depth), float(c_source), _vp(y0_ptr),
                                      _vp(stream or None)))


def debug_math(a, b):
    a = _c(a); b = _c(b)
    out = np.empty((len(a), 9))
    L = load()
    L.pgr_debug_math.restype = ctypes.c_int
    L.pgr_debug_math.argtypes = [_dp, _dp, _i64, _dp]
    check(L.pgr_debug_math(_p(a), _p(b), len(a), _p(out)))
    return out


def device_count():
    return load().pgr_device_count()


def build_info():
    """pgr_build_info(): layout pass applied or not + arithmetic variant of the loaded library."""
    L = load()
    L.pgr_build_info.restype = ctypes.c_char_p
    return L.pgr_build_info().decode()


def device_code_sha256(path=None):
    """sha256 of the gfx950 machine code (.text of the code object inside the library's fat binary): what the
    kernels ARE, independent of symbol order and build paths (two builds of the same source give the same hash).
    profiles/*_traffic.json records it next to the counters; bench.py reports counters only for the code they
    were taken with."""
    import hashlib
    import struct

    def sections(b):
        shoff = struct.unpack_from("<Q", b, 0x28)[0]
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize) for i in range(shnum)]
        stro = secs[shstrndx][4]
        return {b[stro + s_[0]:b.index(b"\0", stro + s_[0])].decode(): (s_[4], s_[5]) for s_ in secs}

    with open(path or LIB_PATH, "rb") as f:
        b = f.read()
    o, n = sections(b)[".hip_fatbin"]
    fb = b[o:o + n]
    if fb[:24] != b"__CLANG_OFFLOAD_BUNDLE__":
        raise PgrError("unexpected fat binary layout")
    cnt = struct.unpack_from("<Q", fb, 24)[0]
    p = 32
    for _ in range(cnt):
        off, size, tl = struct.unpack_from("<QQQ", fb, p)
        p += 24
        triple = fb[p:p + tl].decode()
        p += tl
        if "gfx950" in triple:
            elf = fb[off:off + size]
            to, tn = sections(elf)[".text"]
            return hashlib.sha256(elf[to:to + tn]).hexdigest()
    raise PgrError("no gfx950 code object in the library")


def arrival_histogram_device(device, t_ptr, t_stride, status_ptr, status_stride, n, t_min, t_max, nbins,
                             counts_ptr, stream=0):
    """pgr_arrival_histogram_device on raw device pointers (ints); see include/pgr.h."""
    L = load()
    L.pgr_arrival_histogram_device.restype = ctypes.c_int
    L.pgr_arrival_histogram_device.argtypes = [ctypes.c_int, _vp, _i64, _vp, _i64, _i64, ctypes.c_double,
                                               ctypes.c_double, ctypes.c_int32, _vp, _vp]
    check(L.pgr_arrival_histogram_device(int(device), _vp(t_ptr), int(t_stride), _vp(status_ptr or None),
                                         int(status_stride), int(n), float(t_min), float(t_max), int(nbins),
                                         _vp(counts_ptr), _vp(stream or None)))
