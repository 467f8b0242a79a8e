"""pr.shoot_rays on configs[2] (range-dependent tables, 1e5 rays, 1000 km, S = 1001), eager -- ONE counted call for the PMC
passes of scripts/collect_profiles_r05.sh (WRITE_SIZE / FETCH_SIZE of the fan kernel and of the un-blocking pass).
usage: api_cfg2_run.py [blocked|rows]"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import pygenray_amd as pr
from pygenray_amd.launch_rays import _device_env
mode = sys.argv[1] if len(sys.argv) > 1 else "blocked"
env_rd, _ = bench.munk_tables(1000e3, nr=101, sofar_slope=2e-4)
h, _ = _device_env(env_rd, False, False, 0)
h.set_option("api_blocked", 1 if mode == "blocked" else 0)
angles = np.linspace(-20, 20, 100_000)
pr.shoot_rays(1000.0, 0.0, angles[:1000], 1000e3, 1001, env_rd, debug=False, flatearth=False)
t0 = time.perf_counter()
fan = pr.shoot_rays(1000.0, 0.0, angles, 1000e3, 1001, env_rd, debug=False, flatearth=False, device_resident=False)
dt = time.perf_counter() - t0
print(json.dumps({"mode": mode, "wall_ms": dt * 1e3, "rays_kept": len(fan), "sample_bytes": int(fan.ts.nbytes + fan.zs.nbytes + fan.ps.nbytes)}))
