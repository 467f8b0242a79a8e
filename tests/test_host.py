"""Host-side logic that needs no GPU: the environment front end, result containers, the
C-ABI library's exported symbols, loud failure without the extension, and the multi-GPU
sharding / all-gather path under gloo (world_size 2, CPU)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import pygenray_amd as pr
from pygenray_amd import OceanEnvironment2D, DataArray, Ray, RayFan, munk_ssp, eflat, eflatinv
from pygenray_amd.environment import _unpack_envi, _mirror_envi_arrays

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- environment (REF tests/test_environment.py)
def test_munk_ssp():
    z = np.arange(0, 6000, 1)
    c = munk_ssp(z, sofar_depth=1300.0)
    assert c.shape == z.shape and z[np.argmin(c)] == pytest.approx(1300.0, abs=2.0)
    assert munk_ssp(np.array([1300.0]))[0] == pytest.approx(1500.0, abs=5.0)


def test_environment_defaults_and_attributes():
    env = OceanEnvironment2D()
    for attr in ("sound_speed", "bathymetry", "dcdz", "bottom_angle", "bottom_angle_interp",
                 "sound_speed_fe", "bathymetry_fe"):
        assert hasattr(env, attr)
    assert env.sound_speed.ndim == 2 and set(env.sound_speed.dims) == {"range", "depth"}
    env2 = OceanEnvironment2D(flat_earth_transform=False)
    assert not hasattr(env2, "sound_speed_fe") and not hasattr(env2, "bathymetry_fe")
    # Q11: the default bathymetry is a 4500 -> 4900 m slope; Q10: slope angle in degrees
    assert env.bathymetry.values[0] == 4500 and env.bathymetry.values[-1] == 4900
    assert env.bottom_angle[0] == pytest.approx(np.degrees(np.arctan(400 / 100e3)))
    assert float(env.bottom_angle_interp(50e3)) == pytest.approx(env.bottom_angle[50], rel=1e-6)


def test_environment_custom_and_validation():
    z = np.arange(0.0, 3000.0, 10.0)
    ssp1 = DataArray(munk_ssp(z), dims=["depth"], coords={"depth": z})
    bathy = DataArray(np.ones(20) * 4000.0, dims=["range"], coords={"range": np.linspace(0, 50e3, 20)})
    env = OceanEnvironment2D(sound_speed=ssp1, bathymetry=bathy, flat_earth_transform=False)
    assert env.sound_speed.ndim == 1
    np.testing.assert_array_equal(env.bathymetry.values, np.ones(20) * 4000.0)
    with pytest.raises(TypeError):
        OceanEnvironment2D(sound_speed=np.ones(100))
    with pytest.raises(TypeError):
        OceanEnvironment2D(bathymetry=np.ones(50))
    with pytest.raises(ValueError):
        OceanEnvironment2D(sound_speed=DataArray(np.ones((5, 10, 20)), dims=["range", "depth", "extra"]))
    with pytest.raises(ValueError):
        OceanEnvironment2D(sound_speed=DataArray(np.ones(50), dims=["range"], coords={"range": np.arange(50)}))
    with pytest.raises(ValueError):
        OceanEnvironment2D(sound_speed=DataArray(np.ones((10, 20)), dims=["depth", "extra"],
                                                 coords={"depth": np.arange(10), "extra": np.arange(20)}))
    with pytest.raises(ValueError):
        OceanEnvironment2D(bathymetry=DataArray(np.ones(50), dims=["depth"], coords={"depth": np.arange(50)}))


def test_eflat_roundtrip():
    dep = np.array([100.0, 500.0, 1000.0, 2000.0, 4000.0])
    cs = np.array([1500.0, 1490.0, 1480.0, 1510.0, 1520.0])
    depf, csf = eflat(dep, 35.0, cs)
    assert np.all(depf > dep)
    dep_rec, cs_rec = eflatinv(depf, np.array([35.0]), csf)
    np.testing.assert_allclose(dep_rec, dep, atol=1e-3)
    np.testing.assert_allclose(cs_rec, cs, rtol=1e-6)


def test_unpack_matches_reference_tables(golden_dir):
    """_unpack_envi reproduces the 7 arrays the reference's _unpack_envi produced (golden g1:
    Munk nz=400; g5: flat-earth tables with the default sloping bathymetry)."""
    g = np.load(os.path.join(golden_dir, "g1_fixture_case.npz"))
    z = np.linspace(0.0, 6000.0, 400)
    r = np.linspace(0.0, 50e3, 30)
    env = OceanEnvironment2D(DataArray(np.outer(np.ones(30), munk_ssp(z)), dims=["range", "depth"],
                                       coords={"range": r, "depth": z}),
                             DataArray(np.full(30, 5000.0), dims=["range"], coords={"range": r}),
                             flat_earth_transform=False)
    got = _unpack_envi(env, flatearth=False)
    for a, k in zip(got, ["cin", "cpin", "rin", "zin", "depths", "depth_ranges", "bottom_angles"]):
        np.testing.assert_array_equal(a, g["env_" + k])
    g = np.load(os.path.join(golden_dir, "g5_flatearth.npz"))
    zf = np.arange(0, 6000, 4.0)
    rf = np.linspace(0.0, 100e3, 100)
    env = OceanEnvironment2D(DataArray(np.outer(np.ones(100), munk_ssp(zf)), dims=["range", "depth"],
                                       coords={"range": rf, "depth": zf}),
                             DataArray(np.linspace(4500, 4900, 100), dims=["range"], coords={"range": rf}),
                             lat=35.0, flat_earth_transform=True)
    got = _unpack_envi(env, flatearth=True)
    for a, k in zip(got, ["cin", "cpin", "rin", "zin", "depths", "depth_ranges", "bottom_angles"]):
        np.testing.assert_allclose(a, g["env_" + k], rtol=1e-15, atol=0)
    # depth-major input is normalised to (range, depth)
    env_t = OceanEnvironment2D(DataArray(np.outer(munk_ssp(zf), np.ones(100)), dims=["depth", "range"],
                                         coords={"range": rf, "depth": zf}), flat_earth_transform=False)
    assert _unpack_envi(env_t, flatearth=False)[0].shape == (100, len(zf))
    with pytest.raises(Exception, match="Flat earth"):
        _unpack_envi(OceanEnvironment2D(flat_earth_transform=False), flatearth=True)


def test_mirror_arrays():
    cin = np.arange(12.0).reshape(3, 4)
    out = _mirror_envi_arrays(cin, cin + 1, np.array([0.0, 1, 3]), np.array([5.0, 6, 7]),
                              np.array([0.0, 2, 3]), np.array([1.0, -2, 3]))
    np.testing.assert_array_equal(out[0], cin[::-1])
    np.testing.assert_array_equal(out[2], [-3.0, -1, 0])
    np.testing.assert_array_equal(out[3], [7.0, 6, 5])
    np.testing.assert_array_equal(out[4], [-3.0, -2, 0])
    np.testing.assert_array_equal(out[5], [-3.0, 2, -1])


# ---------------------------------------------------------------- containers (REF tests/test_ray_objects.py)
def _make_rays(M=3, N=10, R=10000.0):
    rays = []
    for i in range(M):
        r = np.linspace(0.0, R, N)
        theta = float(-5 + i * 5)
        y = np.vstack([r / 1500.0, np.linspace(100.0 + i * 50, 200.0 + i * 50, N),
                       np.ones(N) * np.sin(np.radians(abs(theta) + 1e-3)) / 1500.0])
        rays.append(Ray(r=r, y=y, n_bottom=i % 2, n_surface=0, launch_angle=theta, source_depth=100.0 + i * 50))
    return rays


def test_ray_sign_convention_and_optionals():
    r = np.linspace(0, 1e4, 10)
    y = np.vstack([r / 1500.0, np.linspace(100, 200, 10), np.ones(10) * 1e-4])
    ray = Ray(r=r, y=y, n_bottom=3, n_surface=1, launch_angle=-15.0, source_depth=250.0)
    np.testing.assert_array_equal(ray.z, -y[1])
    np.testing.assert_array_equal(ray.p, -y[2])
    assert (ray.n_bottom, ray.n_surface, ray.launch_angle, ray.source_depth) == (3, 1, -15.0, 250.0)
    bare = Ray(r=r, y=y, n_bottom=0, n_surface=0)
    assert not hasattr(bare, "launch_angle") and not hasattr(bare, "source_depth")


def test_rayfan_shapes_slicing_add_mat(tmp_path):
    rf = RayFan(_make_rays())
    assert rf.thetas.shape == (3,) and rf.rs.shape == rf.ts.shape == rf.zs.shape == rf.ps.shape == (3, 10)
    assert len(rf) == 3 and rf.ray_ids.shape == (3,)
    assert list(rf.ray_ids) == ["-0.0", "0.0b", "0.0"]
    one = rf[1]
    assert isinstance(one, Ray) and one.launch_angle == 0.0
    np.testing.assert_array_equal(one.z, rf.zs[1])  # sign convention survives indexing (Q3)
    assert isinstance(rf[-1], Ray)
    with pytest.raises(IndexError):
        rf[3]
    sub = rf[0:2]
    assert isinstance(sub, RayFan) and len(sub) == 2
    np.testing.assert_array_equal(sub.zs, rf.zs[0:2])
    assert len(rf[np.array([True, False, True])]) == 2 and len(rf[[0, 2]]) == 2
    both = rf + RayFan(_make_rays(M=2))
    assert len(both) == 5
    np.testing.assert_array_equal(both.rs[0], rf.rs[0])
    np.testing.assert_array_equal(both.zs[:3], rf.zs)
    with pytest.raises(TypeError):
        rf + 1
    with pytest.raises(ValueError):
        rf + RayFan(_make_rays(R=5000.0))
    import scipy.io
    rf.save_mat(str(tmp_path / "fan.mat"))
    m = scipy.io.loadmat(str(tmp_path / "fan.mat"))["rayfan"]
    for key in ("thetas", "xs", "ts", "zs", "ps", "n_botts", "n_surfs", "source_depths"):
        assert key in m.dtype.names
    np.testing.assert_allclose(m["zs"][0, 0], rf.zs)


def test_rayfan_from_arrays_equals_list_constructor():
    rays = _make_rays(M=4)
    a = RayFan(rays)
    b = RayFan.from_arrays(a.thetas, a.rs, a.ts, a.zs, a.ps, a.n_botts, a.n_surfs, a.source_depths)
    for k in ("thetas", "rs", "ts", "zs", "ps", "n_botts", "n_surfs", "source_depths", "ray_ids"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k))


def test_rayfan_pickles_and_deep_copies_like_a_plain_object():
    """The reference's RayFan is a plain object: pickle, copy.deepcopy and a trip to a multiprocessing worker keep every
    attribute (a device-resident fan is fetched first: GPU test test_device_resident_fan_pickles_and_releases)."""
    import copy
    import pickle
    a = RayFan(_make_rays(M=4))
    lazy = RayFan.from_arrays(a.thetas, a.rs, a.ts, a.zs, a.ps, a.n_botts, a.n_surfs, a.source_depths)   # ray ids not built yet
    for fan in (a, lazy):
        for b in (pickle.loads(pickle.dumps(fan)), copy.deepcopy(fan)):
            for k in ("thetas", "rs", "ts", "zs", "ps", "n_botts", "n_surfs", "source_depths", "ray_ids"):
                np.testing.assert_array_equal(getattr(fan, k), getattr(b, k))
            assert not b.device_resident and len(b) == 4 and isinstance(b[1], Ray)
    assert a.to_host() is a
    a.release()                     # no-ops on a host fan
    assert a.zs.shape == (4, 10)


def test_plots_smoke():
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import pyplot as plt
    rf = RayFan(_make_rays())
    plt.figure(); rf.plot_ray_fan(); rf.plot_time_front(); rf.plot_time_front(ray_id=True, include_lines=True)
    rf.plot_depth_v_angle(include_line=True); rf[0].plot()
    OceanEnvironment2D().plot()
    plt.close("all")


# ---------------------------------------------------------------- the C ABI library
def test_library_exports_every_declared_symbol():
    from pygenray_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "pgr.h")).read()
    declared = set(re.findall(r"\b(pgr_[a-z_]+)\s*\(", hdr))
    declared -= {"pgr_shoot_fan_"}
    assert {"pgr_env_create", "pgr_shoot_fan", "pgr_shoot_fan_device", "pgr_last_error"} <= declared
    path = _lib.build()  # hipcc cross-compiles gfx950 without a GPU
    L = ctypes.CDLL(path)
    for sym in sorted(declared):
        assert hasattr(L, sym), sym
    # the code object is gfx950 only
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True)
    if out.returncode == 0 and "amdgcn" in out.stdout:
        assert "gfx950" in out.stdout


def test_no_cpu_fallback_and_oracle_not_imported_by_product(tmp_path):
    """The product never imports oracle/, and fails loudly when the HIP library is missing."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import pygenray_amd, numpy as np\n"
        "from pygenray_amd import _lib\n"
        "assert 'oracle' not in sys.modules\n"
        "_lib.LIB_PATH = %r\n"
        "try:\n"
        "    pygenray_amd.shoot_ray(1000., 0., 1., 1e4, 5, pygenray_amd.OceanEnvironment2D(), debug=False)\n"
        "except _lib.PgrError as e:\n"
        "    assert 'no CPU fallback' in str(e); print('LOUD')\n"
    ) % (ROOT, str(tmp_path / "missing.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "LOUD" in out.stdout, out.stderr
    for fn in os.listdir(os.path.join(ROOT, "pygenray_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "pygenray_amd", fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_arithmetic_mode_is_an_import_time_choice():
    """PGR_ARITH (pygenray_amd/_lib.py): 'reference' by default, 'contracted' selects libpgr_hip_fma.so for the whole process,
    anything else is refused at import; both libraries export the C ABI (no GPU needed for any of this)."""
    import ctypes
    code = "import sys; sys.path.insert(0, %r); import pygenray_amd as pr; from pygenray_amd import _lib; print(pr.ARITHMETIC, _lib.LIB_PATH)" % ROOT
    for mode, lib_name in ((None, "libpgr_hip.so"), ("reference", "libpgr_hip.so"), ("contracted", "libpgr_hip_fma.so")):
        env = {k: v for k, v in os.environ.items() if k != "PGR_ARITH"}
        if mode:
            env["PGR_ARITH"] = mode
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert out.returncode == 0, out.stderr[-800:]
        arith, path = out.stdout.split()
        assert arith == (mode or "reference") and os.path.basename(path) == lib_name
    bad = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PGR_ARITH="fast"))
    assert bad.returncode != 0 and "PGR_ARITH" in bad.stderr
    from pygenray_amd import _lib
    if os.path.exists(_lib.CONTRACTED_LIB):
        L = ctypes.CDLL(_lib.CONTRACTED_LIB)
        for sym in ("pgr_env_create", "pgr_shoot_fan", "pgr_shoot_fan_device", "pgr_fan_launch", "pgr_eigen_refine_depths_fn", "pgr_build_info"):
            getattr(L, sym)
        L.pgr_build_info.restype = ctypes.c_char_p
        assert b"contract" in L.pgr_build_info().lower() or b"fma" in L.pgr_build_info().lower()


# ---------------------------------------------------------------- multi-GPU path on gloo (world_size 2)
_DIST_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle
from helpers import munk_arrays, y0_for
from pygenray_amd.distributed import shoot_fan_sharded, shard_indices, arrival_time_histogram
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
arrs = munk_arrays(60e3, nr=10, z=np.arange(0, 6000, 5.0))
theta = np.linspace(-19, 19, 37)     # odd: shards of 19 and 18 rays
y0 = y0_for(oracle, arrs, 1000.0, 0.0, theta)
def compute(y0_local):               # the oracle stands in for the HIP fan on this CPU-only box
    o = oracle.shoot_fan(*arrs, y0_local, 0.0, 60e3, 2)
    end = np.stack([o["T"][:, -1], o["z"][:, -1], o["p"][:, -1]], 1)
    return (torch.from_numpy(end), torch.from_numpy(o["n_bott"]), torch.from_numpy(o["n_surf"]),
            torch.from_numpy(o["status"]))
end, nb, ns, st = shoot_fan_sharded(compute, y0)
full = oracle.shoot_fan(*arrs, y0, 0.0, 60e3, 2)
ref = np.stack([full["T"][:, -1], full["z"][:, -1], full["p"][:, -1]], 1)
assert end.shape == (37, 3)
assert np.array_equal(end.numpy(), ref, equal_nan=True), "gathered fan is not in launch-angle order"
assert np.array_equal(nb.numpy(), full["n_bott"]) and np.array_equal(ns.numpy(), full["n_surf"])
assert np.array_equal(st.numpy(), full["status"])
# the same through records the kernel packs itself (PGR_PACKED_END): start ... finish
from pygenray_amd.distributed import pack_end_records, start_all_gather_records
mine0 = shard_indices(37, dist.get_rank(), 2)
e_l, nb_l, ns_l, st_l = compute(y0[mine0])
rec = pack_end_records(e_l, nb_l, ns_l, st_l, 19)
g2 = start_all_gather_records(rec, 37)
rec.zero_()                          # the next fan may overwrite the records while they travel
end2, nb2, ns2, st2 = g2.finish()
assert np.array_equal(end2.numpy(), ref, equal_nan=True) and np.array_equal(st2.numpy(), full["status"])
assert np.array_equal(nb2.numpy(), full["n_bott"]) and np.array_equal(ns2.numpy(), full["n_surf"])
mine = shard_indices(37, dist.get_rank(), 2)
h_local = arrival_time_histogram(end[mine, 0], st[mine], 16, 39.0, 41.0, reduce=True)
h_full = arrival_time_histogram(end[:, 0], st, 16, 39.0, 41.0)
assert torch.equal(h_local, h_full) and h_full.sum() > 0
dist.barrier(); dist.destroy_process_group()
print("RANK_OK")
"""


def test_sharded_fan_all_gather_gloo_world2(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(_DIST_WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and "RANK_OK" in o, e[-2000:]



# ---------------------------------------------------------------- the sharded API on gloo (world_size 2)
_SHARD_COMMON = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle
import pygenray_amd as pr
from helpers import munk
from pygenray_amd.distributed import (shoot_rays_sharded, find_eigenrays_sharded, arrival_histogram_sharded,
                                      pack_end_records)
# a steep seamount at 30 km: the rays that hit its face bounce backwards and are DROPPED (status 3) in the middle of
# the fan, so brackets form across the gap (REF/eigenrays.py:65-79 works on the surviving rays, Q12)
z = np.arange(0, 6000, 5.0); r = np.linspace(0, 60e3, 13); br = np.linspace(0, 60e3, 61)
ssp = pr.DataArray(np.tile(munk(z), (len(r), 1)), dims=["range", "depth"], coords={"range": r, "depth": z})
bathy = pr.DataArray(5000 - 2600 * np.exp(-((br - 30e3) / 2.5e3) ** 2), dims=["range"], coords={"range": br})
env = pr.OceanEnvironment2D(ssp, bathy, flat_earth_transform=False)
arrs = pr._unpack_envi(env, flatearth=False)
ZS, X1, S = 1000.0, 60e3, 21
theta = np.linspace(-19, 19, 77)

def compute(y0_local, x0, x1, backwards, n_pad):     # the oracle stands in for the HIP fan on this CPU-only box
    o = oracle.shoot_fan(*arrs, y0_local, x0, x1, 2)
    end = np.stack([o["T"][:, -1], o["z"][:, -1], o["p"][:, -1]], 1)
    return pack_end_records(torch.from_numpy(end), torch.from_numpy(o["n_bott"]), torch.from_numpy(o["n_surf"]),
                            torch.from_numpy(o["status"]), n_pad)

C0 = oracle.bilinear(0.0, ZS, arrs[2], arrs[3], arrs[0])
def shoot1(th_user, S_):
    y0 = np.array([[0.0, ZS, np.sin(np.radians(-th_user)) / C0]])
    return oracle.shoot_fan(*arrs, y0, 0.0, X1, S_)

def refine(z1s, z2s, th1s, th2s, rd):                # REF/eigenrays.py:206-268, bracket by bracket
    n = len(z1s)
    found = np.zeros(n, bool); th = np.zeros(n)
    T = np.zeros((n, S)); Z = np.zeros((n, S)); P = np.zeros((n, S)); nb = np.zeros(n, np.int64); ns = np.zeros(n, np.int64)
    for k in range(n):
        z1, z2, t1, t2 = z1s[k], z2s[k], th1s[k], th2s[k]
        t = t1 - (z1 + rd) * (t2 - t1) / (z2 - z1)
        for it in range(23):
            o = shoot1(t, S)
            if o["status"][0] != 0:
                break
            zr = -o["z"][0, -1]
            if abs(zr + rd) < 1.0:
                found[k] = True; th[k] = t
                T[k], Z[k], P[k], nb[k], ns[k] = o["T"][0], -o["z"][0], -o["p"][0], o["n_bott"][0], o["n_surf"][0]
                break
            if np.sign(zr + rd) == np.sign(z1 + rd): z1, t1 = zr, t
            else: z2, t2 = zr, t
            t = t1 - (z1 + rd) * (t2 - t1) / (z2 - z1)
            if it > 20: break
    return found, th, np.linspace(0.0, X1, S), T, Z, P, nb, ns

def run(tag):
    fan, raw = shoot_rays_sharded(ZS, 0.0, theta, X1, env, flatearth=False, compute=compute, return_all=True)
    er = find_eigenrays_sharded(fan, [1000.0, 2500.0, 70.0], ZS, 0.0, X1, S, env, refine=refine)
    h, edges = arrival_histogram_sharded(ZS, 0.0, theta, X1, env, 32, 39.0, 41.0, flatearth=False, compute=compute)
    out = dict(status=raw["status"], thetas=fan.thetas, ts=fan.ts, zs=fan.zs, ps=fan.ps, nb=fan.n_botts, hist=h, edges=edges)
    for k in (0, 1, 2):
        out[f"e{k}_th"] = er.launch_angles[k]; out[f"e{k}_ts"] = er.ts[k]; out[f"e{k}_zs"] = er.zs[k]
        out[f"e{k}_nb"] = er.n_botts[k]; out[f"e{k}_ra"] = er.received_angles[k]
        out[f"e{k}_failed"] = np.array(er.failed_eray_theta_brackets[k], dtype=float).reshape(-1, 2)
        out[f"e{k}_n"] = np.array([er.num_eigenrays[[1000.0, 2500.0, 70.0][k]], er.num_eigenrays_found[k]])
    np.savez(tag, **out)
"""

_SHARD_WORKER = _SHARD_COMMON + r"""
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=%(world)d)
if %(world)d > 2:
    # no pickled objects on the wire: the eigenray results travel as fixed-shape tensor collectives
    def _no_objects(*a, **k):
        raise AssertionError("all_gather_object used")
    dist.all_gather_object = _no_objects
run(sys.argv[2])
dist.barrier(); dist.destroy_process_group()
print("RANK_OK")
"""

_SHARD_SINGLE = _SHARD_COMMON + r"""
run(sys.argv[1])
print("SINGLE_OK")
"""


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_shoot_eigenrays_histogram_gloo_equals_single_process(tmp_path, world):
    """pygenray_amd.distributed's API (shoot_rays_sharded -> find_eigenrays_sharded, arrival_histogram_sharded) on two
    gloo ranks against the very same calls in one process without torch.distributed: every rank must hold the
    single-process fan (launch order, dropped rays gone), the same EigenRays -- brackets straddle the two ranks' rays
    (with a strided deal EVERY bracket does) and one spans the rays a seamount drops in the middle of the fan --
    and the same histogram.  The oracle stands in for the HIP fan and for the device refinement (CPU-only box).
    world 4: uneven shards (77 rays: 20 / 19 / 19 / 19), a receiver depth with fewer brackets than ranks (ranks with ZERO
    brackets still join both collectives), dropped rays on every rank's shard, and no ``all_gather_object`` anywhere."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    w = tmp_path / "worker.py"; w.write_text(_SHARD_WORKER % dict(root=ROOT, port=port, world=world))
    one = tmp_path / "single.py"; one.write_text(_SHARD_SINGLE % dict(root=ROOT))
    procs = [subprocess.Popen([sys.executable, str(w), str(r), str(tmp_path / f"rank{r}.npz")], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    p1 = subprocess.run([sys.executable, str(one), str(tmp_path / "single.npz")], capture_output=True, text=True, timeout=600)
    assert p1.returncode == 0 and "SINGLE_OK" in p1.stdout, p1.stderr[-2000:]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and "RANK_OK" in o, e[-2000:]
    ref = np.load(tmp_path / "single.npz")
    # the scenario is the one the docstring promises
    st = ref["status"]
    inner = np.where(st != 0)[0]
    assert len(inner) >= 3 and inner.min() > 0 and inner.max() < len(st) - 1          # dropped rays INSIDE the fan
    assert ref["e0_n"][0] >= 4 and ref["e0_n"][1] >= 3 and ref["e1_n"][0] >= 2
    th = ref["thetas"]
    gap = np.where(np.diff(th) > 1.5 * (38 / 76))[0]                                   # the fan's neighbours across the gap
    assert len(gap) >= 1
    spans = [rd for rd in (1000.0, 2500.0) if np.intersect1d(np.where(np.diff(np.sign(ref["zs"][:, -1] + rd)))[0], gap).size]
    assert spans, "no bracket spans the dropped rays"
    assert ref["hist"].sum() > 0.5 * (st == 0).sum()
    if world == 4:
        assert len(st) % 4 != 0 and 0 < ref["e2_n"][0] < 4          # uneven shards; ranks without a bracket at 70 m
        for q in range(4):
            assert (st[q::4] != 0).any()                              # every shard holds dropped rays
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert sorted(got.files) == sorted(ref.files)
        for k in ref.files:
            assert np.array_equal(got[k], ref[k], equal_nan=True), (r, k)


def test_eigenrays_export_and_plots(tmp_path):
    """EigenRays.save_mat (schema of REF/ray_objects.py:604-636), plot / plot_ducted / plot_angle_time
    (REF/ray_objects.py:550-602) on a hand-made result."""
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import pyplot as plt
    import scipy.io
    import pygenray_amd as pr
    from pygenray_amd.ray_objects import EigenRays, RayFan
    z = np.arange(0, 5000, 10.0); r = np.linspace(0, 50e3, 6)
    env = pr.OceanEnvironment2D(pr.DataArray(np.tile(pr.munk_ssp(z), (6, 1)), dims=["range", "depth"], coords={"range": r, "depth": z}),
                                pr.DataArray(np.full(6, 4500.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
    S = 9
    rs = np.tile(np.linspace(0, 50e3, S), (3, 1))
    fan0 = RayFan.from_arrays(np.array([-3.0, 1.0, 8.0]), rs, rs / 1500.0, -np.linspace(1000, 1200, S) * np.ones((3, 1)),
                              np.array([[1e-4], [-1e-4], [2e-4]]) * np.cos(np.linspace(0, 6, S)), np.array([0, 0, 2]), np.array([0, 0, 1]),
                              np.full(3, 1000.0))
    fan1 = fan0[:1]
    er = EigenRays([1000.0, 2000.0], {0: fan0, 1: fan1}, env, {1000.0: 4, 2000.0: 1}, {0: 3, 1: 1}, {0: [(2.0, 2.5)], 1: []})
    assert er.received_angles[0].shape == (3,) and list(er.ray_id[0][:2]) != [] and er.ray_id[0][2].endswith("b")
    path = str(tmp_path / "er.mat")
    er.save_mat(path)
    m = scipy.io.loadmat(path, squeeze_me=True, struct_as_record=False)["eigenrays"]
    d0 = m.receiver_depth_0
    for k in ("receiver_depth", "xs", "ts", "zs", "ps", "received_angles", "launch_angles", "ray_id", "ray_id_int", "n_bottom",
              "n_surface", "source_depth", "num_eigenrays", "num_eigenrays_found"):
        assert hasattr(d0, k), k
    assert d0.receiver_depth == 1000.0 and np.array_equal(d0.zs, er.zs[0]) and np.array_equal(d0.launch_angles, er.launch_angles[0])
    assert np.array_equal(d0.n_bottom, [0, 0, 2]) and m.receiver_depth_1.xs.shape == (S,)
    for f in (lambda: er.plot(), lambda: er.plot(0, c="r"), lambda: er.plot([0, 1]), er.plot_ducted, lambda: er.plot_ducted(lw=2),
              er.plot_angle_time, lambda: er.plot_angle_time([1])):
        plt.figure(); f(); plt.close("all")
    # plot_ducted draws only the rays that touched nothing, depth positive down
    plt.figure(); er.plot_ducted()
    lines = plt.gca().get_lines()
    assert len(lines) == 2 + 1 and np.all(lines[0].get_ydata() > 0)
    plt.close("all")


def test_flat_earth_c_and_range_dependent_transform():
    """flat_earth_c / OceanEnvironment2D.flat_earth_transform_rd (REF/environment.py:156-173, 239-303): column by column
    eflat at the column's latitude, interpolated back onto the original depth grid (what xarray's interp does)."""
    import scipy.interpolate
    import pygenray_amd as pr
    z = np.arange(0, 5500, 10.0); r = np.linspace(0, 200e3, 7); lat = np.linspace(20, 50, 7)
    cv = np.array([pr.munk_ssp(z, 1300 + 1e-4 * ri) for ri in r])
    for dims, vals in ((["range", "depth"], cv), (["depth", "range"], cv.T)):
        c = pr.DataArray(vals, dims=dims, coords={"range": r, "depth": z, "lat": lat})
        f = pr.flat_earth_c(c)
        assert tuple(f.dims) == ("range", "depth") and f.values.shape == (7, len(z))
        for i in (0, 3, 6):
            depf, cf = pr.eflat(z, lat[i], cv[i])
            want = scipy.interpolate.interp1d(depf, cf, bounds_error=False)(z)
            np.testing.assert_allclose(f.values[i], want, rtol=1e-14, equal_nan=True)
        assert np.all(f.values[:, 1:] > cv[:, 1:])          # the flattened speeds are larger at depth
    env = pr.OceanEnvironment2D(c.transpose("range", "depth"), pr.DataArray(np.full(7, 5000.0), dims=["range"], coords={"range": r}),
                                flat_earth_transform=False)
    env.flat_earth_transform_rd()
    assert np.array_equal(env.sound_speed_fe.values, f.values) and np.array_equal(env.bathymetry_fe.values, env.bathymetry.values)
    arrs = pr._unpack_envi(env, flatearth=True)            # the 7-array contract takes it
    assert arrs[0].shape == (7, len(z)) and np.array_equal(arrs[3], z)
    with pytest.raises(ValueError):
        pr.flat_earth_c(pr.DataArray(cv, dims=["range", "depth"], coords={"range": r, "depth": z}))

def test_pack_and_interleave_single_process():
    import torch
    from pygenray_amd.distributed import pack_end_records, all_gather_fan, shard_indices
    end = torch.arange(15, dtype=torch.float64).reshape(5, 3)
    nb = torch.tensor([1, 2, 3, 4, 5], dtype=torch.int32)
    buf = pack_end_records(end, nb, nb * 2, nb * 0, 6)
    assert buf.shape == (6, 5) and buf[5].abs().sum() == 0
    e2, b2, s2, st2 = all_gather_fan(end, nb, nb * 2, nb * 0, 5)
    assert torch.equal(e2, end) and torch.equal(b2, nb) and torch.equal(s2, nb * 2)
    assert list(shard_indices(10, 1, 4)) == [1, 5, 9]


def _build_c_example(tmp_path):
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "pygenray_amd", "csrc")
    if not os.path.exists(os.path.join(lib_dir, "libpgr_hip.so")) or shutil.which("gcc") is None:
        pytest.skip("libpgr_hip.so or gcc not available")
    exe = os.path.join(str(tmp_path), "shoot_fan")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "shoot_fan.c"), "-o", exe, "-L" + lib_dir, "-lpgr_hip",
                           "-lm", "-Wl,-rpath," + lib_dir])
    return exe


def test_header_is_plain_c_and_example_links(tmp_path):
    """include/pgr.h is valid C99 and every entry point the plain-C example uses resolves against
    libpgr_hip.so (compile + link only: no GPU needed)."""
    assert os.path.exists(_build_c_example(tmp_path))


def test_committed_bench_line_and_profiles_are_well_formed():
    """profiles/ carries the round's rocprofv3 summaries and the bench line they belong to."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r02_bench_line.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "roofline_valu", "cpu_baseline",
              "cpu_baseline_c", "eigenray", "build"):
        assert k in d, k
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is not None and 0 < d["roofline_valu"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1
    assert "solve_ivp" in d["cpu_baseline"]["sample"] and d["build"].startswith("layout: relaid")
    for k in ("wall_s", "fan_s", "search_s", "brackets", "found", "failed", "launches"):
        assert k in d["eigenray"], k
    stats = open(os.path.join(root, "profiles", "r02_kernel_stats.csv")).read()
    assert "pgr_fan_kernel" in stats
    tr = json.load(open(os.path.join(root, "profiles", "r02_traffic.json")))
    assert tr["sample"]["hbm_gb_per_launch"] > 0 and tr["sample"]["valu_wave_instructions_per_launch"] > 1e9
    c2 = json.load(open(os.path.join(root, "profiles", "r02_config2_counters.json")))
    assert 0.9 < c2["end_state_only"]["l2_hit_rate"] <= 1 and "pgr_fan_kernel<false" in open(
        os.path.join(root, "profiles", "r02_config2_kernel_stats.csv")).read()



def test_round3_profiles_name_the_binary_and_bench_line_carries_the_legs():
    """profiles/r03_*: the PMC passes of the three kernels name the binary they describe (sha256 of the gfx950 machine
    code, pgr_build_info(), git commit) and the committed bench line carries the legs, the lone-wave floor and counters
    that belong to the code it ran."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tr = json.load(open(os.path.join(root, "profiles", "r03_traffic.json")))
    assert len(tr["device_code_sha256"]) == 64 and tr["build"].startswith("layout: relaid") and len(tr["git_commit"]) == 40
    for k in ("sample", "sample-nosave", "flatearth-sample", "flatearth-sample-nosave", "rangedep-sample", "rangedep-sample-nosave"):
        assert tr[k]["rays"] == 100000 and tr[k]["hbm_gb_per_launch"] > 0, k
    assert tr["sample"]["valu_wave_instructions_per_launch"] > 1e9 and tr["flatearth-sample"]["valu_wave_instructions_per_launch"] > 1e9
    for v, tag in (("", "<true, 4, 1>"), ("_flatearth", "<true, 5, 1>"), ("_rangedep", "<false, 4, 1>")):
        assert "pgr_fan_kernel" + tag in open(os.path.join(root, "profiles", f"r03_kernel_stats{v}.csv")).read(), v
    d = json.load(open(os.path.join(root, "profiles", "r03_bench_line.json")))
    for k in ("legs", "lone_wave_ms", "device_code_sha256", "roofline", "cpu_baseline", "cpu_baseline_c", "eigenray"):
        assert k in d, k
    assert set(d["legs"]) == {"flatearth_default", "range_dependent", "rays_1e6"}
    for leg in ("flatearth_default", "range_dependent"):
        for mode in ("end_state", "trajectories"):
            assert d["legs"][leg][mode]["kernel_ms"] > 0 and 0 < d["legs"][leg][mode]["frac"] < 1
    assert d["legs"]["flatearth_default"]["end_state"]["kernel_ms"] < 1.2 * d["lone_wave_ms"]["end_state"] * 1.1
    assert d["cpu_baseline"]["jit"] is False and d["metric"].startswith("ray-steps/sec (whole node), 1e5-ray Munk fan")
    # counters are reported only for the code they were taken with
    if d["roofline"]["traffic"] is not None:
        assert d["device_code_sha256"] == tr["device_code_sha256"]
    else:
        assert "device_code_sha256" in d["roofline"]["traffic_source"] or "no PMC pass" in d["roofline"]["traffic_source"]


def test_round4_profiles_and_bench_line():
    """profiles/r04_*: PMC passes that name their binary (incl. the sample-blocked configs[2] kernel and its halved write
    traffic), the instruction budget, the all-rays S = 1001 parity log, and the committed bench line with the api leg."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tr = json.load(open(os.path.join(root, "profiles", "r04_traffic.json")))
    assert len(tr["device_code_sha256"]) == 64 and tr["build"].startswith("layout: relaid")
    assert tr["rangedep-blocked"]["WRITE_SIZE_KB"] * 1024 / 1e9 <= 2.9 < tr["rangedep-sample"]["WRITE_SIZE_KB"] * 1024 / 1e9
    for v, tag in (("", "<true, 4, 1>"), ("_flatearth", "<true, 5, 1>"), ("_rangedep", "<false, 4, 1>"), ("_rangedep_blocked", "<false, 4, 3>")):
        assert "pgr_fan_kernel" + tag in open(os.path.join(root, "profiles", f"r04_kernel_stats{v}.csv")).read(), v
    b = json.load(open(os.path.join(root, "profiles", "r04_isa_budget.json")))
    assert b["device_code_sha256"] == tr["device_code_sha256"]
    pt = b["derived"]["headline"]["per_wave_trip"]
    assert 600 < pt["SQ_INSTS_VALU"] < 800 and 11.9 < pt["SQ_INSTS_LDS"] < 12.1 and pt["lane_utilisation"] > 0.98
    log = open(os.path.join(root, "profiles", "r04_bitparity_S1001.txt")).read()
    assert log.count("all 1001 samples (SciPy order) 1.00000") == 9 and log.count("bit-equal True") >= 3 and "0.99" not in log.split("default sample")[0]
    d = json.load(open(os.path.join(root, "profiles", "r04_bench_line.json")))
    assert d["device_code_sha256"] == tr["device_code_sha256"] and d["roofline"]["traffic"] is not None
    assert set(d["legs"]) == {"flatearth_default", "range_dependent", "rays_1e6", "api"}
    rd = d["legs"]["range_dependent"]
    assert rd["trajectories"]["kernel_ms"] < rd["trajectories_row_layout"]["kernel_ms"]
    api = d["legs"]["api"]
    assert api["device_resident"]["wall_ms"] < api["eager"]["wall_ms"] and api["eager"]["trajectory_bytes_to_host"] > 2e9
    reh = json.load(open(os.path.join(root, "profiles", "r04_bench_line_4ranks_one_gpu_rehearsal.json")))
    assert reh["ranks_joined"] == 4 and reh["legs"]["config4"]["gathered_rays"] == 4_000_000
    assert reh["legs"]["config4"]["histogram_counted_rays"] == reh["legs"]["config4"]["gathered_ok"]


def test_round5_profiles_and_bench_line():
    """profiles/r05_*: ONE binary -- every file that names a device_code_sha256 names the one the committed bench line was
    produced with --, kernel stats for the persistent 1e6-ray instance and the lone steepest wave, the API's blocked-kernel
    write traffic, the all-rays parity logs, the bench line's new legs and the five-rank rehearsal's per-GPU roofline."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    P = os.path.join(root, "profiles")
    tr = json.load(open(os.path.join(P, "r05_traffic.json")))
    sha = tr["device_code_sha256"]
    assert len(sha) == 64 and tr["build"].startswith("layout: relaid")
    named = 0
    for f in sorted(glob.glob(os.path.join(P, "r05_*"))):
        txt = open(f).read()
        if f.endswith(".json"):
            d = json.loads(txt)
            d = d[0] if isinstance(d, list) else d
            for key in ("device_code_sha256", "device_code_sha256_of_the_product_it_was_built_beside"):
                if key in d:
                    assert d[key] == sha, (os.path.basename(f), key)
                    named += 1
        else:
            for m in re.finditer(r"device_code_sha256 ([0-9a-f]{64})", txt):
                assert m.group(1) == sha, os.path.basename(f)
                named += 1
    assert named >= 12
    for v, tags in (("", ["<true, 4, 1, false>"]), ("_flatearth", ["<true, 5, 1, false>"]), ("_rangedep", ["<false, 4, 1, false>"]),
                    ("_rangedep_blocked", ["<false, 4, 3, false>"]), ("_1e6", ["<true, 4, 0, true>"]),
                    ("_lone_wave", ["<true, 4, 1, false>", "<true, 4, 0, false>"]), ("_api_config2", ["<false, 4, 3, false>", "pgr_unblock_cols"])):
        txt = open(os.path.join(P, f"r05_kernel_stats{v}.csv")).read()
        for t in tags:
            assert ("pgr_fan_kernel" + t if t.startswith("<") else t) in txt, (v, t)
    # the 1e6-ray leg: rocprofv3's average of the persistent instance agrees with the bench line's HIP events
    row = [ln for ln in open(os.path.join(P, "r05_kernel_stats_1e6.csv")) if "<true, 4, 0, true>" in ln][0].split('",')
    avg_ms = float(row[1].split(",")[2]) / 1e6
    d = json.load(open(os.path.join(P, "r05_bench_line.json")))
    leg = d["legs"]["rays_1e6"]
    assert abs(avg_ms - leg["end_state"]["kernel_ms"]) < 0.03 * avg_ms
    assert leg["end_state"]["frac"] >= 0.40 and leg["end_state"]["kernel_ms"] <= 36.5 and leg["trajectories"]["frac"] >= 0.40
    assert d["device_code_sha256"] == sha and d["roofline"]["traffic"] is not None and 0.30 < d["roofline"]["frac"] < 0.34
    assert set(d["legs"]) == {"flatearth_default", "range_dependent", "rays_1e6", "api", "api_config2", "fma_contracted"}
    fma = d["legs"]["fma_contracted"]
    assert "FMA contraction" in fma["build"] and fma["device_code_sha256"] != sha and "NOT the reference's arithmetic" in fma["note"]
    assert fma["trajectories"]["kernel_ms"] < d["roofline"]["kernel_ms"]
    assert all(g["rays_beyond_the_10x_rule"] == 0 and g["bounce_counts_equal"] for g in fma["against_the_reference"].values())
    api2 = tr["api-config2"]
    assert api2["blocked"]["fan_kernel_write_over_sample_bytes"] <= 1.25 < 2.0 < api2["rows"]["fan_kernel_write_over_sample_bytes"]
    assert tr["sample-nosave@1000000"]["rays"] == 1_000_000
    sq = tr["sample-nosave@1000000-sq_counters"]
    assert sq["SQ_WAVES"] == 2048.0        # persistent: 256 workgroups x 8 waves, whatever the ray count
    wt = json.load(open(os.path.join(P, "r05_wave_times.json")))
    st, pe = wt["rays_1e6_static_deal_of_whole_workgroups"], wt["rays_1e6_persistent_waves"]
    assert st["slot_time_split"]["idle_inside_a_resident_workgroup"] > 0.15 and pe["simd_residency"]["two_or_more_waves"] > 0.93
    svc = json.load(open(os.path.join(P, "r05_service_times.json")))
    assert 15e3 < svc["end_state"]["cycles_per_service"] < 25e3 and len(svc["end_state"]["sections"]) == 18
    log = open(os.path.join(P, "r05_bitparity_S1001.txt")).read()
    assert log.count("all 1001 samples (SciPy order) 1.00000") == 9 and log.count("bit-equal True") >= 3 and "0.99" not in log.split("default sample")[0]
    log6 = open(os.path.join(P, "r05_bitparity_1e6_rays.txt")).read()
    assert "rays 1000000" in log6 and log6.count("all 11 samples (SciPy order) 1.00000") == 3 and "n=999535" in log6
    b = json.load(open(os.path.join(P, "r05_isa_budget.json")))
    assert b["device_code_sha256"] == sha and 600 < b["derived"]["headline"]["per_wave_trip"]["SQ_INSTS_VALU"] < 800
    reh = json.load(open(os.path.join(P, "r05_bench_line_5ranks_one_gpu_rehearsal.json")))
    c4 = reh["legs"]["config4"]
    assert reh["ranks_joined"] == 5 and c4["gathered_rays"] == 5_000_000 and c4["histogram_counted_rays"] == c4["gathered_ok"]
    assert len(c4["roofline_per_gpu"]["ranks"]) == 5 and all(q["kernel_ms"] > 0 and q["frac"] > 0 for q in c4["roofline_per_gpu"]["ranks"])
    assert reh["eigenray_sharded"]["fan_rays_per_gpu"] == 200_000 and "REHEARSAL" in reh["config"]["sharding"]


def test_round6_profiles_and_bench_line():
    """profiles/r06_*: the committed bench line, the rocprofv3 summaries and the PMC passes name ONE binary; the line carries the
    legs round 6 added (headline end state, per-step kernel times); rocprofv3's per-call durations agree with the line's HIP
    events once the cold warm-up calls are left out; the 1e6-ray trajectory instance writes ~1.08 x its samples."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    P = os.path.join(root, "profiles")
    tr = json.load(open(os.path.join(P, "r06_traffic.json")))
    d = json.load(open(os.path.join(P, "r06_bench_line.json")))
    sha = tr["device_code_sha256"]
    assert len(sha) == 64 and d["device_code_sha256"] == sha and tr["build"].startswith("layout: relaid")
    for name in ("config2", "config2_blocked", "flatearth", "rays_1e6", "rays_1e6_traj"):
        x = json.load(open(os.path.join(P, f"r06_bench_line_{name}.json")))
        assert x["device_code_sha256"] == sha, name
    for f in ("r06_fuzz_sweeps.txt", "r06_bitparity_S1001.txt", "r06_bitparity_1e6_rays.txt", "r06_final_checks.txt"):
        found = re.findall(r"device_code_sha256 ([0-9a-f]{64})", open(os.path.join(P, f)).read())
        assert found and all(v == sha for v in found), f
    # ALL rays of the three benched workloads (x 1001 samples) and ALL 1e6 rays of the configs[3] / [4] fan, this round's library
    bp = open(os.path.join(P, "r06_bitparity_S1001.txt")).read()
    assert bp.count("status equal: True") == 3 and len(re.findall(r"all ok\s+n=\s*\d+\s+bit-equal: end state 1\.00000  n_steps 1\.00000  n_rej 1\.00000  bounces 1\.00000  all 1001 samples \(SciPy order\) 1\.00000", bp)) == 3
    assert bp.count("end states bit-equal True; step / bounce counts equal True") == 3 and "equal False" not in bp and "equal: False" not in bp
    b6 = open(os.path.join(P, "r06_bitparity_1e6_rays.txt")).read()
    assert "rays 1000000" in b6 and re.search(r"all ok\s+n=999535\s+bit-equal: end state 1\.00000  n_steps 1\.00000  n_rej 1\.00000  bounces 1\.00000", b6) and "equal: False" not in b6
    # the driver's line: metric / workload of BASELINE.json, roofline + cpu_baseline, the legs
    assert d["metric"] == json.load(open(os.path.join(root, "BASELINE.json")))["metric"] and d["dtype"] == "f64" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.30 < r["frac"] < 0.34 and r["traffic"] is not None and 2.4 < r["traffic_gb_per_launch"] < 2.8
    each = r["kernel_ms_each"]
    assert len(each) == d["steps"] == 20 and abs(sum(each) / len(each) - r["kernel_ms"]) < 1e-3
    assert r["kernel_ms"] <= d["ms_per_step"] <= 1.02 * r["kernel_ms"]
    assert abs(d["value"] - d["config"]["ray_steps_per_pass"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert set(d["legs"]) == {"flatearth_default", "range_dependent", "rays_1e6", "headline_end_state", "api", "api_config2", "fma_contracted"}
    he = d["legs"]["headline_end_state"]
    assert he["bytes_per_ray_step"] == 80.0 and he["kernel_ms"] < r["kernel_ms"] and he["ray_steps"] == d["config"]["ray_steps_per_pass"]
    assert d["legs"]["rays_1e6"]["end_state"]["frac"] >= 0.40 and d["legs"]["rays_1e6"]["trajectories"]["frac"] >= 0.40
    assert d["lone_wave_ms"]["trajectories"] < r["kernel_ms"]          # the fan cannot beat its steepest wave
    # rocprofv3 of the same commands: the instances, and per-call durations that agree with the HIP events once warm
    for v, tag in (("", "<true, 4, 1, false>"), ("_end_state", "<true, 4, 0, false>"), ("_flatearth", "<true, 5, 1, false>"),
                   ("_rangedep", "<false, 4, 1, false>"), ("_rangedep_blocked", "<false, 4, 3, false>"), ("_1e6", "<true, 4, 0, true>"),
                   ("_1e6_traj", "<true, 4, 1, true>"), ("_api_config2", "<false, 4, 3, false>")):
        assert "pgr_fan_kernel" + tag in open(os.path.join(P, f"r06_kernel_stats{v}.csv")).read(), v
    kc = json.load(open(os.path.join(P, "r06_kernel_calls.json")))
    head = [e for e in kc["stats"] if "<true, 4, 1, false>" in e["kernel"]][0]
    assert head["n_calls"] == 12 and head["stats_mean_ms"] >= head["warm_mean_ms"] and abs(head["warm_mean_ms"] - r["kernel_ms"]) < 0.015 * r["kernel_ms"]
    big = [e for e in kc["1e6traj_stats"] if "<true, 4, 1, true>" in e["kernel"]][0]
    x = json.load(open(os.path.join(P, "r06_bench_line_rays_1e6_traj.json")))
    assert abs(big["warm_mean_ms"] - x["roofline"]["kernel_ms"]) < 0.02 * big["warm_mean_ms"] and x["roofline"]["frac"] >= 0.42
    assert x["roofline"]["traffic"] is not None and x["roofline"]["kernel"].startswith("pgr_fan_kernel<true, 4, 1, true>")
    t6 = tr["sample@1000000"]
    assert t6["rays"] == 1000000 and 1.0 < t6["WRITE_SIZE_KB"] * 1024 / (1e6 * 1001 * 24) < 1.15
    # the narrow-wave proxy's record (verdict r5 item 1): baseline, lone packets, re-dealt fans, every run bit-equal
    txt = open(os.path.join(P, "r06_narrow_wave_proxy.txt")).read()
    assert "baseline save=1" in txt and "lone steepest packet, 16 rays" in txt and txt.count("end states bit-equal True") >= 20 and "bit-equal False" not in txt


def test_eval_cache_fingerprint_is_cheap_and_sees_in_place_edits():
    """host_physics._fingerprint (the content check of the tables kept on the device for point-by-point derivsrd / event
    calls): whole-array hashes up to 1 MB, a strided sample beyond -- an in-place edit of a whole table, of a row, of a
    column or of its ends changes it; a large table costs well under a millisecond per query."""
    import time
    from pygenray_amd.host_physics import _fingerprint
    rng = np.random.default_rng(3)
    small = [rng.normal(size=(20, 300)), rng.normal(size=(20, 300)), np.arange(20.0), np.arange(300.0), np.ones(20), np.arange(20.0)]
    f0 = _fingerprint(small)
    small[0][7, 123] += 1e-9
    f1 = _fingerprint(small)
    assert f0 != f1 and _fingerprint(small) == f1            # a single element of a small table
    big = [rng.normal(size=(400, 6000)), rng.normal(size=(400, 6000)), np.arange(400.0), np.arange(6000.0), np.ones(400), np.arange(400.0)]
    f0 = _fingerprint(big)
    for edit in (lambda a: a.__iadd__(1e-9), lambda a: a[17].__iadd__(1e-9), lambda a: a[:, 4321].__iadd__(1e-9),
                 lambda a: a[-1:, -1:].__iadd__(1e-9), lambda a: a[:1, :1].__iadd__(1e-9)):
        edit(big[0])
        f1 = _fingerprint(big)
        assert f1 != f0
        f0 = f1
    # whole-row and whole-column edits for shapes whose flat sampling step shares a factor with the row length
    # (300 x 6000: step 27, gcd 3 -- the round-5 sample never visited column 4321), few wide rows, many narrow rows,
    # a 3-D table, a non-contiguous (transposed) view
    shapes = [(300, 6000), (101, 5001), (3, 200001), (200001, 3), (150000, 10), (50, 40, 300)]
    for shp in shapes:
        a = rng.normal(size=shp)
        ncols, nrows = shp[-1], int(np.prod(shp[:-1]))
        m = a.reshape(nrows, ncols)
        g0 = _fingerprint([a])
        for col in sorted({1, ncols // 3, (4321 % ncols), ncols - 2}):
            m[:, col] += 1e-9
            g1 = _fingerprint([a])
            assert g1 != g0, (shp, "column", col)
            g0 = g1
        for row in sorted({1, nrows // 3, nrows - 2, 2 % nrows}):
            m[row, :] += 1e-9
            g1 = _fingerprint([a])
            assert g1 != g0, (shp, "row", row)
            g0 = g1
    at = rng.normal(size=(6000, 300)).T            # (300, 6000), Fortran order
    g0 = _fingerprint([at])
    at[:, 4321] += 1e-9
    assert _fingerprint([at]) != g0
    t0 = time.perf_counter()
    for _ in range(20):
        _fingerprint(big)
    assert (time.perf_counter() - t0) / 20 < 5e-3           # (19 MB tables: was ~40 ms per query with whole-array crc32)
    assert _fingerprint([np.arange(5, dtype=np.int32)]) == _fingerprint([np.arange(5.0)])     # content, not dtype, for small arrays


def test_device_code_hash_reads_the_built_library():
    """_lib.device_code_sha256: the gfx950 .text inside the library's fat binary (no GPU needed)."""
    from pygenray_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libpgr_hip.so not built")
    h = _lib.device_code_sha256()
    assert len(h) == 64 and h == _lib.device_code_sha256(_lib.LIB_PATH)

# ----------------------------------------------------------------------------- build: instruction layout
def test_instruction_layout_pass_plans_encodings():
    """pygenray_amd/_isa_layout.py: 4-byte e32 VALU instructions are re-encoded as 8-byte e64 ones
    exactly where that keeps 8-byte instructions off the 32-byte fetch-window boundaries."""
    from pygenray_amd import _isa_layout as L
    # a 4-byte instruction followed by a long run of 8-byte ones: every 4th straddles ...
    items = [("i", 4, True)] + [("i", 8, False)] * 40
    promote, before, after = L.plan_function(items)
    assert before == 10 and after == 0 and promote == {0}
    # ... and nothing can be done when the 4-byte instruction has no 8-byte encoding
    items = [("i", 4, False)] + [("i", 8, False)] * 40
    promote, before, after = L.plan_function(items)
    assert before == 10 and after == 10 and not promote
    # an alignment directive restarts the window
    items = [("i", 4, False), ("a", 6, 0)] + [("i", 8, False)] * 8
    assert L.plan_function(items)[1:] == (0, 0)
    # already clean layouts are left alone
    items = [("i", 8, False), ("i", 4, True), ("i", 4, True)] * 20
    promote, before, after = L.plan_function(items)
    assert before == 0 and after == 0 and not promote
    # which instructions have an e64 twin: plain register / inline-constant operands only
    assert L.promotable("v_fmac_f64_e32 v[0:1], v[2:3], v[4:5]")
    assert L.promotable("v_mov_b32_e32 v1, s5")
    assert L.promotable("v_cndmask_b32_e32 v1, v2, v3, vcc")
    assert L.promotable("v_cmp_lt_f64_e32 vcc, 0, v[2:3]")
    assert L.promotable("v_add_u32_e32 v1, -1, v2")
    assert not L.promotable("v_mov_b32_e32 v1, 0x3ff00000")      # literal: no VOP3 form on gfx9
    assert not L.promotable("v_add_u32_e32 v1, 1000, v2")
    assert not L.promotable("v_addc_co_u32_e32 v1, vcc, v2, v3, vcc")
    assert not L.promotable("v_mul_f64 v[0:1], v[2:3], v[4:5]")  # already 8 bytes
    assert not L.promotable("s_mov_b32 s0, s1")


def test_build_reports_that_the_layout_pass_was_applied():
    """The instruction-layout pass used to fail silently (plain hipcc build, one stderr line): now the
    library itself says what it is (pgr_build_info), bench.py prints it, and this test requires the
    product library in the tree to be the re-encoded one."""
    import ctypes
    from pygenray_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libpgr_hip.so not built")
    L = ctypes.CDLL(_lib.LIB_PATH)     # loads without a GPU; no HIP call is made
    L.pgr_build_info.restype = ctypes.c_char_p
    info = L.pgr_build_info().decode()
    assert info.startswith("layout: relaid, "), info
    m = __import__("re").search(r"relaid, (\d+) -> (\d+) straddling", info)
    assert m and int(m.group(2)) < 0.3 * int(m.group(1)), info
    assert "correctly rounded div/sqrt/pow/asin/sin" in info and "NOT bit-identical" not in info


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus N` run plainly launches N ranks (fresh children, one per GPU) before
    touching a GPU; rank 0 reports; a wrong external WORLD_SIZE is an error, not a silent 1-GPU run.
    Rehearsed on CPU with gloo (--launcher-only: the ranks join, all-reduce a counter, exit)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-only", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and json.loads(line[0]) == {"launcher_only": True, "n_gpus": 2, "ranks_joined": 2}
    env2 = dict(env, WORLD_SIZE="3", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-only", "--backend", "gloo"],
                       env=env2, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE 3" in r.stderr
    # the driver's 8-GPU shape: eight ranks start, join and report (the one-GPU box may hold at most six processes on its
    # card, so the eight-rank rehearsal of the launcher is this CPU one; six ranks share the GPU in profiles/r05_*6ranks*)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--launcher-only", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and json.loads(line[0]) == {"launcher_only": True, "n_gpus": 8, "ranks_joined": 8}


def test_real_xarray_and_xr_lite_unpack_to_the_same_arrays():
    """pygenray takes xarray.DataArray inputs (REF/environment.py:6,49-119); this package ships a
    minimal stand-in (xr_lite) so that it runs where xarray is absent, and accepts the real thing.
    Where xarray is importable both must unpack to identical arrays (differentiate = np.gradient,
    edge_order 1; REF/launch_rays.py:717-742)."""
    xr = pytest.importorskip("xarray")
    import pygenray_amd as pr
    from pygenray_amd.environment import _unpack_envi
    z = np.arange(0, 6000, 5.0)
    r = np.linspace(0, 200e3, 21)
    c = np.array([pr.munk_ssp(z, 1300 + 2e-4 * ri) for ri in r])
    b = 4500 + 300 * np.sin(r / 40e3)
    for fe in (False, True):
        envs = []
        for DA in (pr.DataArray, xr.DataArray):
            envs.append(pr.OceanEnvironment2D(DA(c, dims=["range", "depth"], coords={"range": r, "depth": z}),
                                              DA(b, dims=["range"], coords={"range": r}), flat_earth_transform=fe))
        for a, bb in zip(_unpack_envi(envs[0], flatearth=fe), _unpack_envi(envs[1], flatearth=fe)):
            assert np.array_equal(np.asarray(a), np.asarray(bb))
