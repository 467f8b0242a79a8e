"""What bit-identity costs and what it buys: a library variant built with FMA contraction allowed (-DPGR_FMA
-ffp-contract=fast: the compiler fuses a*b + c, the reciprocal square root is the 2-ulp Newton form) timed on the headline
fan, and its end states against the REFERENCE's own vectors g11 / g12 / g13 (tests/golden, rule B of tests/helpers.py: the
deviation over the reference's own 7-sample self-noise).  NOT the product arithmetic: the product reproduces the
reference's operation order bit for bit; this tells what that choice is worth.  Prints one JSON object.
usage (GPU box): python scripts/fma_leg.py --lib pygenray_amd/csrc/libpgr_hip_fma.so"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--passes", type=int, default=6)
a = ap.parse_args()
from pygenray_amd import _lib
_lib.LIB_PATH = os.path.abspath(a.lib)
import torch
from helpers import load, tiled_env, munk_arrays, REL_TOL, NOISE_FACTOR
from pygenray_amd.device_fan import DeviceFan, fan_y0

out = {"library": os.path.relpath(_lib.LIB_PATH, ROOT), "build": _lib.build_info(), "device_code_sha256": _lib.device_code_sha256(),
       "note": "FMA contraction allowed: NOT the reference's arithmetic, no parity credit -- reported so that what the bit-identical "
               "default costs (kernel time) and buys (deviation from the reference against the reference's own noise) are both in the line"}
# ---- the headline fan
arrs = munk_arrays(1000e3)
env = _lib.EnvHandle(*arrs)
y0 = fan_y0(arrs, 1000.0, 0.0, -np.linspace(-20, 20, 100_000))
for name, save in (("trajectories", True), ("end_state", False)):
    fan = DeviceFan(env, y0, 0.0, 1000e3, 1001, save=save, sample_major=True)
    for _ in range(3):
        fan.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.passes):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fan.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    steps = fan.ray_steps()
    alive = max(fan.N - int((fan.status != 0).sum().item()), 1)
    b_alg = 80.0 + (24.0 * 1001 / (steps / alive) if save else 0.0)
    ms = float(np.mean(ts))
    out[name] = {"kernel_ms": ms, "kernel_ms_min": float(np.min(ts)), "ray_steps": steps, "bytes_per_ray_step": b_alg,
                 "frac": steps * b_alg / (ms * 1e-3) / 1e9 / 8000.0}
    del fan
env.close()
# ---- the reference's vectors: end states against the reference's own self-noise (rule B, not asserted here)
gold = {}
for tag, fname, arrs_of, x1 in (("g11 configs[1] x 288", "g11_munk_1000km_288.npz", tiled_env, 1000e3),
                                ("g12 configs[2] x 128", "g12_config2_128.npz",
                                 lambda g: munk_arrays(float(g["r_max"]), nr=int(g["nr"]), sofar_slope=float(g["sofar_slope"])), 1000e3),
                                ("g13 default environment 1000 km x 32", "g13_default_env_1000km.npz", tiled_env, 1000e3)):
    g = load(fname)
    e = _lib.EnvHandle(*arrs_of(g))
    o = e.shoot_fan(g["y0"], 0.0, x1, 101, exact_samples=True)
    e.close()
    ok = g["ok"].astype(bool)
    same_fate = bool(np.array_equal(o["status"] == 0, ok))
    both = ok & (o["status"] == 0)
    end = np.stack([o["T"][:, -1], o["z"][:, -1], o["p"][:, -1]], 1)[both]
    gend = np.stack([g["T"][:, -1], g["z"][:, -1], g["p"][:, -1]], 1)[both]
    scale = np.array([float(np.nanmax(g["T"][ok])), float(g["zin"][-1]) if "zin" in g.files else 6000.0, 1 / 1500.0])
    d = np.abs(end - gend)
    sn = g["selfnoise_end"][both]
    needs = d > REL_TOL * scale
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(needs, d / sn, 0.0)
    gold[tag] = {"rays": int(both.sum()), "same_rays_dropped_as_the_reference": same_fate,
                 "bounce_counts_equal": bool(np.array_equal(o["n_bott"][both], g["n_bott"][both]) and np.array_equal(o["n_surf"][both], g["n_surf"][both])),
                 "rays_beyond_1e-8": int(needs.any(1).sum()), "worst_ratio_to_reference_self_noise": float(np.nanmax(ratio)) if needs.any() else 0.0,
                 "rays_beyond_the_10x_rule": int((d > np.maximum(NOISE_FACTOR * sn, REL_TOL * scale)).any(1).sum()),
                 "worst_rel_T_z_p": [float(v) for v in (d / scale).max(0)]}
out["against_the_reference"] = gold
print(json.dumps(out))
