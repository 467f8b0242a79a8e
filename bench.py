#!/usr/bin/env python3
"""Headline benchmark: ray-steps/sec of a 1e5-ray Munk fan to 1000 km (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one fan resident in HBM: per rank, 100 000 rays
(weak scaling: the global fan has N x 100 000 launch angles linspace(-20, 20), dealt to the
ranks in a strided fashion), Munk profile z = arange(0, 6000, 1), 100 range columns to
1000 km, flat bottom 5000 m, source (0 m, 1000 m), rtol 1e-9, 1001 saved samples per ray
(the trajectories pygenray's RayFan holds).  With N > 1 each step ends with the RCCL
all-gather of the 40-byte end records (pygenray_amd/distributed.py).

One JSON line on rank 0: value = accepted RK45 steps of all rays on all ranks / wall time
(max over ranks, barrier + synchronize on both sides).  `roofline` prices the fan kernel
against HBM with SURVEY.md 8(d)'s algorithmic bytes; `cpu_baseline` times the CPU oracle
(a port of the reference's integrator -- the reference itself cannot travel to the GPU box)
on a bounded sample of the same workload, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAYS_PER_GPU = 100_000
RANGE_M = 1000e3
S_SAVE = 1001
SOURCE_DEPTH = 1000.0
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def munk_tables(r_max, nr=100):
    import pygenray_amd as pr
    z = np.arange(0, 6000, 1.0)
    r = np.linspace(0.0, r_max, nr)
    ssp = pr.DataArray(np.tile(pr.munk_ssp(z), (nr, 1)), dims=["range", "depth"],
                       coords={"range": r, "depth": z})
    bathy = pr.DataArray(np.full(nr, 5000.0), dims=["range"], coords={"range": r})
    env = pr.OceanEnvironment2D(ssp, bathy, flat_earth_transform=False)
    from pygenray_amd.environment import _unpack_envi
    return env, _unpack_envi(env, flatearth=False)


def host_cores():
    """Cores this process may really use: min(affinity, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(arrs, n_rays=25000):
    """CPU oracle (C port of the reference integrator, OpenMP over rays, one thread per usable
    core) on a bounded sample: every (100000/n_rays)-th ray of the same fan, full 1000 km,
    trajectories included (about 30 core-seconds)."""
    import oracle
    from pygenray_amd.device_fan import fan_y0
    theta = np.linspace(-20, 20, RAYS_PER_GPU)[:: RAYS_PER_GPU // n_rays][:n_rays]
    y0 = fan_y0(arrs, SOURCE_DEPTH, 0.0, -theta)
    oracle.lib()
    cores = host_cores()
    oracle.set_num_threads(cores)
    t0 = time.time()
    out = oracle.shoot_fan(*arrs, y0, 0.0, RANGE_M, S_SAVE)
    dt = time.time() - t0
    steps = int(out["n_steps"].sum())
    return {"value": steps / dt, "unit": "ray-steps/s", "cores": cores,
            "kind": "port",
            "sample": f"{len(y0)} rays (every {RAYS_PER_GPU // n_rays}th of the 1e5-ray fan), "
                      f"1000 km, {steps} ray-steps in {dt:.1f} s, oracle/ray_oracle.c with OpenMP, "
                      f"{cores} threads (cgroup CPU quota of the box)"}


def scipy_baseline(arrs, n_rays=12):
    """The NumPy/SciPy call pattern of the reference (solve_ivp RK45 + events), one core."""
    from oracle import scipy_port
    from pygenray_amd.device_fan import fan_y0
    theta = np.linspace(-20, 20, RAYS_PER_GPU)[:: RAYS_PER_GPU // n_rays][:n_rays]
    y0 = fan_y0(arrs, SOURCE_DEPTH, 0.0, -theta)
    t0 = time.time()
    out = scipy_port.shoot_fan(*arrs, y0, 0.0, RANGE_M, S_SAVE)
    dt = time.time() - t0
    steps = int(out["n_steps"].sum())
    return {"value": steps / dt, "unit": "ray-steps/s", "cores": 1, "kind": "port",
            "sample": f"{len(y0)} rays, 1000 km, {steps} ray-steps in {dt:.1f} s, "
                      f"oracle/scipy_port.py (scipy.integrate.solve_ivp, un-jitted RHS)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU, help="rays per GPU")
    ap.add_argument("--layout", choices=["ray", "sample"], default="sample",
                    help="trajectory layout in HBM: [S][N] (default: coalesced stores; the drop-in API\n"
                         "hands RayFan a transposed (N,S) view of it) or [N][S] (4.4x HBM write amplification)")
    ap.add_argument("--no-save", action="store_true", help="end state only (B_alg = 80 B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--waves-per-block", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pygenray_amd import _lib
    from pygenray_amd.device_fan import DeviceFan, fan_y0
    from pygenray_amd.distributed import shard_indices, all_gather_fan, start_all_gather_records

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0 and world > 1:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    # PGR_BENCH_FORCE_DIST=1 rehearses the N > 1 code path (RCCL init, all-gather, reductions)
    # with a single rank on a one-GPU box
    use_dist = world > 1 or os.environ.get("PGR_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    _lib.load()
    if args.waves_per_block:
        _lib.set_waves_per_block(args.waves_per_block)
    env_obj, arrs = munk_tables(RANGE_M)
    env = _lib.EnvHandle(*arrs, device=local_rank)
    n_global = args.rays * world
    theta = np.linspace(-20, 20, n_global)
    idx = shard_indices(n_global, rank, world)
    y0 = fan_y0(arrs, SOURCE_DEPTH, 0.0, -theta[idx])  # ODE angle = -user (>= 70-ray branch)
    save = not args.no_save
    # N > 1: the kernel writes the 40-byte end records of the all-gather itself (PGR_PACKED_END)
    fan = DeviceFan(env, y0, 0.0, RANGE_M, S_SAVE, save=save, sample_major=(args.layout == "sample"),
                    packed_end=use_dist, n_pad=(n_global + world - 1) // world)

    def step():
        fan.run()
        if use_dist:
            return start_all_gather_records(fan.records, n_global).finish()
        return None

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    pending = None
    for k in range(args.steps):
        ev[k][0].record()
        fan.run()
        ev[k][1].record()
        if use_dist:
            # the end records of pass k travel (RCCL stream) while pass k+1 integrates; every
            # gathered fan is reassembled in launch-angle order before the clock stops
            started = start_all_gather_records(fan.records, n_global)
            if pending is not None:
                pending.finish()
            pending = started
    if pending is not None:
        pending.finish()
    fence()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    local_steps = fan.ray_steps()
    n_drop = int((fan.status != 0).sum().item())
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ss = torch.tensor([local_steps, n_drop], dtype=torch.int64, device="cuda")
        dist.all_reduce(ss, op=dist.ReduceOp.SUM)
        total_steps, n_drop = int(ss[0].item()), int(ss[1].item())
    else:
        total_steps = local_steps

    if rank == 0:
        value = total_steps * args.steps / dt
        # B_alg = 80 B (state in + out) + 24 B per saved (T,z,p) sample
        mean_steps = local_steps / max(fan.N - int((fan.status != 0).sum().item()), 1)
        b_alg = 80.0 + (24.0 * S_SAVE / mean_steps if save else 0.0)
        achieved = local_steps * b_alg / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (profiles/r01_traffic.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE)
        traffic = traffic_gb = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.layout}{'' if save else '-nosave'}"
                if key in tj and tj[key].get("rays") == fan.N:
                    traffic_gb = tj[key]["hbm_gb_per_launch"]
                    traffic = traffic_gb / (kern_ms * 1e-3)
            except Exception:
                traffic = None
        out = {
            "metric": "ray-steps/sec (whole node), 1e5-ray Munk fan to 1000 km",
            "value": value, "unit": "ray-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: Munk SSP dz=1 m, 100000 launch angles per GPU "
                                   "linspace(-20,20), 1000 km, rtol 1e-9, fp64",
                       "rays_per_gpu": fan.N, "num_range_save": S_SAVE if save else 0,
                       "trajectory_layout": args.layout if save else "none",
                       "ray_steps_per_pass": total_steps, "dropped_rays": n_drop,
                       "sharding": "strided launch angles, all-gather of end records" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_gb_per_launch": traffic_gb,
                         "algorithmic_gb_per_launch": local_steps * b_alg / 1e9,
                         "kernel": f"pgr_fan_kernel<true, 4, {1 if save else 0}> (table in LDS, zin = j * 1 m, "
                                   f"{'linspace save grid' if save else 'end state only'})", "kernel_ms": kern_ms,
                         "bytes_per_ray_step": b_alg,
                         "note": "algorithmic bytes per SURVEY 8(d); the stepper keeps state in "
                                 "VGPRs and the SSP table in LDS, so it is fp64-VALU bound, not HBM bound"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(arrs)
            out["cpu_baseline_scipy"] = scipy_baseline(arrs)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
