"""PGR_ARITH=contracted -- the supported opt-in arithmetic (pygenray_amd/_lib.py): libpgr_hip_fma.so, the same sources
with FMA contraction allowed.  One library per process, chosen at import, so this module has two halves:

* ``test_contracted_mode_in_its_own_process`` (runs in the DEFAULT mode): starts a child interpreter with
  PGR_ARITH=contracted that runs the second half of this module and the drop-in API tests (tests/test_dropin_api.py) there.
* the ``contracted_*`` tests (run only in a PGR_ARITH=contracted process): the loaded library is the contracted one, and
  its fans meet rule (B) of tests/helpers.py -- within 1e-8 x scale or 10 x the self-noise of the REFERENCE's own vectors,
  class medians within 1e-8 -- on g2 ... g13.  NO bit-parity claim is made or tested for this mode: the reference
  arithmetic (the default) is the only one the parity statements of DESIGN.md are about.

Why the mode exists: pygenray jits its physics with ``fastmath=True`` (REF/integration_processes.py:26,101,177) and SciPy's
stage sums run through BLAS dot products, so FMA contraction is inside the envelope of the reference's own arithmetic.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import load, env_from, tiled_env, munk_arrays

pytestmark = pytest.mark.gpu
# rule (B)'s self-noise for this mode: the oracle under +-1, 2, 3-ulp perturbations of p0 (+ rtol +-1 ulp) -- the sampling the
# reference's own self-noise was recorded with in the golden vectors (seven end states per ray).  Contracted arithmetic is one
# more draw from that noise: against a spread estimated from +-1 ulp alone (four runs) 2 of g11's 288 rays exceeded 10 x.
NOISE_ULPS = (1, 2, 3)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _arith():
    from pygenray_amd import _lib
    return _lib.ARITH


def test_contracted_mode_in_its_own_process():
    from pygenray_amd import _lib
    if _lib.ARITH != "reference":
        pytest.skip("this IS the contracted process")
    if not os.path.exists(_lib.CONTRACTED_LIB):
        pytest.fail("libpgr_hip_fma.so is not built (__graft_entry__.build() builds it beside the product)")
    env = dict(os.environ, PGR_ARITH="contracted", PGR_EIGEN_STRICT="0")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_contracted_arith.py"), os.path.join(ROOT, "tests", "test_dropin_api.py")],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-3000:] + out.stderr[-1500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "CONTRACTED_LIBRARY_LOADED" in out.stdout, tail
    print(out.stdout[-600:])


@pytest.fixture(scope="module")
def clib():
    from pygenray_amd import _lib
    if _lib.ARITH != "contracted":
        pytest.skip("needs PGR_ARITH=contracted at import (test_contracted_mode_in_its_own_process starts that process)")
    _lib.load()
    return _lib


def test_contracted_library_is_the_one_loaded(clib, capsys):
    import pygenray_amd as pr
    assert pr.ARITHMETIC == "contracted" and clib.LIB_PATH == clib.CONTRACTED_LIB
    info = clib.build_info()
    assert "contract" in info.lower() or "fma" in info.lower(), info
    # the product library stays untouched beside it, and the two hold different device code
    assert clib.device_code_sha256(clib.CONTRACTED_LIB) != clib.device_code_sha256(clib.REFERENCE_LIB)
    with capsys.disabled():
        print("\nCONTRACTED_LIBRARY_LOADED:", info)


def test_contracted_meets_rule_b_on_the_reference_vectors(clib):
    """Rule (B) only (bit_parity=False, noise_ulps=NOISE_ULPS): the contracted fan against the vectors the reference itself produced."""
    from test_hip_parity import golden_check
    g = load("g2_munk_100km.npz")
    golden_check(clib, g, tiled_env(g), 0.0, 100e3, 101, label="g2 [contracted]", bit_parity=False, noise_ulps=NOISE_ULPS)
    g = load("g3_munk_1000km.npz")
    golden_check(clib, g, tiled_env(g), 0.0, 1000e3, 101, label="g3 [contracted]", bit_parity=False, noise_ulps=NOISE_ULPS)
    g = load("g9_irregular_grids.npz")
    golden_check(clib, g, env_from(g), 1e3, 69e3, 61, prefix="t9_", rtol=1e-9, label="g9 [contracted]", strict_bouncing=False,
                 bit_parity=False, noise_ulps=NOISE_ULPS)
    g = load("g4_range_dependent.npz")
    floor = dict(T=1e-6, z=1e-2, p=1e-7)
    golden_check(clib, g, env_from(g), 10e3, 90e3, 81, prefix="fwd_", label="g4 fwd [contracted]", abs_floor=floor,
                 strict_bouncing=False, bit_parity=False, noise_ulps=NOISE_ULPS)
    g = load("g5_const_c.npz")
    golden_check(clib, g, env_from(g), 0.0, 30e3, 60, label="const c [contracted]", bit_parity=False, noise_ulps=NOISE_ULPS)
    g = load("g5_flatearth.npz")
    golden_check(clib, g, env_from(g), 0.0, 100e3, 101, label="flat earth [contracted]", strict_bouncing=False, bit_parity=False, noise_ulps=NOISE_ULPS)


def test_contracted_meets_rule_b_at_the_headline_range(clib):
    from test_hip_parity import golden_check
    from test_oracle_golden import end_state_check
    g = load("g11_munk_1000km_288.npz")
    out = golden_check(clib, g, tiled_env(g), 0.0, 1000e3, 101, label="g11 [contracted]", bit_parity=False, noise_ulps=NOISE_ULPS)
    print("g11 end states [contracted]:", end_state_check(g, out, "g11"))
    g = load("g12_config2_128.npz")
    arrs = munk_arrays(float(g["r_max"]), nr=int(g["nr"]), sofar_slope=float(g["sofar_slope"]))
    out = golden_check(clib, g, arrs, 0.0, 1000e3, 101, label="g12 [contracted]", bit_parity=False, noise_ulps=NOISE_ULPS)
    print("g12 end states [contracted]:", end_state_check(g, out, "g12"))
    for tag, x1 in (("100km", 100e3), ("1000km", 1000e3)):
        g = load(f"g13_default_env_{tag}.npz")
        out = golden_check(clib, g, tiled_env(g), 0.0, x1, 101, label="g13 " + tag + " [contracted]", bit_parity=False, noise_ulps=NOISE_ULPS)
        print(f"g13 {tag} end states [contracted]:", end_state_check(g, out, "g13 " + tag))


def test_contracted_reference_fixture_through_dropin_api(clib):
    """The reference's committed regression fixture (its own tolerances) through the drop-in API in contracted mode."""
    import pygenray_amd as pr
    z = np.linspace(0.0, 6000.0, 400)
    r = np.linspace(0.0, 50e3, 30)
    ssp = pr.DataArray(np.outer(np.ones(30), pr.munk_ssp(z)), dims=["range", "depth"], coords={"range": r, "depth": z})
    bathy = pr.DataArray(np.full(30, 5000.0), dims=["range"], coords={"range": r})
    env = pr.OceanEnvironment2D(sound_speed=ssp, bathymetry=bathy, flat_earth_transform=False)
    rf = pr.shoot_rays(1300.0, 0.0, [-8.0, -4.0, 0.0, 4.0, 8.0], 50e3, 50, env, n_processes=1, debug=False, flatearth=False)
    ref = load("ref_munk_regression.npz")
    np.testing.assert_allclose(rf.ts, ref["ts"], atol=5e-6)
    np.testing.assert_allclose(rf.zs, ref["zs"], atol=0.1)
    np.testing.assert_allclose(rf.ps, ref["ps"], atol=0.1)
    np.testing.assert_array_equal(rf.n_botts, ref["n_botts"])
    np.testing.assert_array_equal(rf.n_surfs, ref["n_surfs"])
