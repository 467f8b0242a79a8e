"""SURVEY section 5, "race detection / sanitizers": the CPU side of the repository -- the C restatement of the
reference's integrator (oracle/ray_oracle.c, `make -C oracle asan`) and the host twin of the device's correctly
rounded functions (tests/crmath_host.c) -- built with -fsanitize=address,undefined and run once through the
golden suite / the correctly-rounded checks in a child python that has libasan preloaded.  CPU only: the GPU
pool offers no sanitizer (and the product has no CPU path to sanitize)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _run_under_asan(args, extra_env):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan here")
    env = dict(os.environ)
    env.update(extra_env)
    env["LD_PRELOAD"] = asan
    # (python itself is not instrumented: its arenas look like leaks; everything else aborts the child)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["OMP_NUM_THREADS"] = "4"
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
    return out


def test_oracle_golden_suite_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    out = _run_under_asan(["tests/test_oracle_golden.py"], {"ORACLE_SANITIZE": "1"})
    assert " passed" in out and "failed" not in out
    # the child really loaded the sanitised build
    probe = subprocess.run([sys.executable, "-c", "import oracle, sys; oracle.lib(); "
                            "print(any('_asan/libray_oracle.so' in l for l in open('/proc/self/maps')))"],
                           cwd=ROOT, capture_output=True, text=True,
                           env=dict(os.environ, ORACLE_SANITIZE="1", LD_PRELOAD=_runtime("libasan.so"),
                                    ASAN_OPTIONS="detect_leaks=0"))
    assert probe.stdout.strip().endswith("True"), probe.stdout + probe.stderr


def test_crmath_host_twin_under_asan_ubsan():
    out = _run_under_asan(["tests/test_crmath.py"], {"CRMATH_SANITIZE": "1", "CRMATH_N": "200000"})
    assert " passed" in out and "failed" not in out
