"""Device and host memory over many calls of the drop-in API (eager and device-resident fans, trajectory fetches, eigenray
searches, environments created and dropped): nothing may grow."""
import sys, os, gc, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pygenray_amd as pr


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


def rss_mb():
    return int(open("/proc/self/statm").read().split()[1]) * 4096 / 2**20


z = np.arange(0, 6000, 1.0); r = np.linspace(0, 300e3, 60)
# argv[2] = "rd": a range-dependent environment (tables in HBM / L2: the API's sample-blocked kernel + un-blocking pass)
RD = len(sys.argv) > 2 and sys.argv[2] == "rd"
ssp = (lambda: np.array([pr.munk_ssp(z, 1300.0 + 2e-4 * ri) for ri in r])) if RD else (lambda: np.tile(pr.munk_ssp(z), (60, 1)))
mk = lambda: pr.OceanEnvironment2D(pr.DataArray(ssp(), dims=["range", "depth"], coords={"range": r, "depth": z}),
                                   pr.DataArray(np.full(60, 5000.0), dims=["range"], coords={"range": r}), flat_earth_transform=False)
ang = np.linspace(-18, 18, 20_000)
env = mk()
log = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    fan = pr.shoot_rays(1000.0, 0.0, ang, 300e3, 201, env, debug=False, flatearth=False, device_resident=bool(it % 2))
    if it % 3 == 0:
        _ = fan.zs[0, -1]
    if it % 10 == 0:
        er = pr.find_eigenrays(fan, [500.0, 1000.0], 1000.0, 0.0, 300e3, 51, env, debug=False, flatearth=False, quiet=True)
        del er
    if it % 7 == 0:
        env = mk()          # the old environment (tables, workspaces, pooled fan buffers) goes with its last fan
    del fan
    gc.collect()
    log.append((free_mb(), rss_mb()))
    if it in (9, 19, 39, 59) or it % 40 == 39:
        print(f"call {it + 1:3d}: device free {log[-1][0]:9.0f} MB   host RSS {log[-1][1]:7.0f} MB", flush=True)
d_dev = log[19][0] - log[-1][0]
d_rss = log[-1][1] - log[19][1]
print(f"calls 20 -> 60: device memory in use grew by {d_dev:.0f} MB, host RSS by {d_rss:.0f} MB")
assert d_dev < 64 and d_rss < 256, "memory grows with the number of calls"
print("leak check OK")
