"""How many host cores does the CPU baseline really get?  (affinity, cgroup quota, measured scaling)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
import oracle
from helpers import munk_arrays
from pygenray_amd.device_fan import fan_y0
arrs = munk_arrays(1000e3)
theta = np.linspace(-20, 20, 100000)[::16]
y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
oracle.lib()
for nt in (1, 4, 8, 16, 32, 64, 128):
    os.environ["OMP_NUM_THREADS"] = str(nt)
    try: oracle.set_num_threads(nt)
    except Exception as e: print("no set_num_threads", e); break
    n = len(y0) if nt > 1 else 200
    t0 = time.time(); out = oracle.shoot_fan(*arrs, y0[:n], 0.0, 1000e3, 1001); dt = time.time() - t0
    print(nt, "threads", n, "rays", f"{out['n_steps'].sum()/dt:.3e} ray-steps/s", flush=True)
