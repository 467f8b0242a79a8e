#!/usr/bin/env python3
"""Headline benchmark: ray-steps/sec of a 1e5-ray Munk fan to 1000 km (BASELINE.json configs[1]),
plus the eigenray wall-clock of configs[3].

    python bench.py [--gpus N] [--steps K] [--warmup W]

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES the N ranks (one per
GPU, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`,
before anything here touches a GPU) and exits with their status; under an external torchrun it is a
rank.  A "step" is one pass of the hot path over one fan resident in HBM: per rank, `--rays` rays
(default 100 000; weak scaling: the global fan has N x rays launch angles linspace(-20, 20), dealt
to the ranks in a strided fashion), Munk profile z = arange(0, 6000, 1), 100 range columns to
1000 km, flat bottom 5000 m, source (0 m, 1000 m), rtol 1e-9, 1001 saved samples per ray (the
trajectories pygenray's RayFan holds; `--no-save`: end state only).  With N > 1 each step ends with
the RCCL all-gather of the 40-byte end records (pygenray_amd/distributed.py); `--histogram` adds the
4096-bin arrival-time histogram of configs[4] (HIP kernel + all-reduce of the counts), so that
`--rays 1000000 --no-save --histogram` is configs[4]'s per-GPU shape.

One JSON line on rank 0: value = accepted RK45 steps of all rays on all ranks / wall time (max over
ranks, barrier + synchronize on both sides).  `roofline` prices the fan kernel against HBM with
SURVEY.md 8(d)'s algorithmic bytes, `roofline_valu` against the fp64 VALU issue rate that really
bounds it; `cpu_baseline` is the reference's own call pattern (scipy.integrate.solve_ivp, RK45,
4 terminal events, NumPy right-hand side: oracle/scipy_port.py -- the reference itself cannot
travel to the GPU box) on all usable host cores, `cpu_baseline_c` the C/OpenMP oracle; `eigenray`
the wall-clock of configs[3] (1e6-angle fan + regula-falsi refinement); `legs` = the kernel instances users hit
beside the headline one (default flat-earth grid, configs[2], 1e6 rays), a few passes each, and `lone_wave_ms` the
64 steepest rays alone (the floor of the fan's kernel time); rank 0 at N = 1 only.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAYS_PER_GPU = 100_000
BASELINE_METRIC = "ray-steps/sec (whole node), 1e5-ray Munk fan to 1000 km; eigenray wall-clock"   # BASELINE.json
RANGE_M = 1000e3
S_SAVE = 1001
SOURCE_DEPTH = 1000.0
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SIMDS = 256 * 4            # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9           # max shader clock (MI355X_MICROARCH.md)
FP64_CYCLES_PER_WAVE_INSTR = 4  # 16 fp64 lanes per clock per SIMD: a wave64 instruction holds the pipe 4 cycles


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU, help="rays per GPU")
    ap.add_argument("--layout", choices=["ray", "sample"], default="sample",
                    help="trajectory layout in HBM: [S][N] (default: coalesced stores; the drop-in API\n"
                         "hands RayFan a transposed (N,S) view of it) or [N][S] (4.4x HBM write amplification)")
    ap.add_argument("--no-save", action="store_true", help="end state only (B_alg = 80 B)")
    ap.add_argument("--histogram", action="store_true",
                    help="each step also bins the arrival times (4096 bins) on the device and all-reduces the counts")
    ap.add_argument("--range-dependent", action="store_true",
                    help="BASELINE configs[2]: sofar axis sloping 2e-4 over 101 range columns (tables stay in HBM/L2); "
                         "not the headline workload")
    ap.add_argument("--flat-earth", action="store_true",
                    help="the configs[1] tables after OceanEnvironment2D's default flat-earth transform (non-uniform zin, "
                         "kernel ZM = 5): what pr.shoot_rays(...) integrates with the reference's default arguments; not the headline workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eigenray", action="store_true")
    ap.add_argument("--blocked", action="store_true",
                    help="(with --range-dependent) trajectories in the sample-blocked layout [S/4][N][4] (PGR_SAMPLE_BLOCKED): what the "
                         "range_dependent leg's `trajectories` entry times")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra kernel legs (flat-earth default grid, configs[2], 1e6 rays, lone wave)")
    ap.add_argument("--eigen-rays", type=int, default=1_000_000, help="fan size of the eigenray leg (configs[3])")
    ap.add_argument("--waves-per-block", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; tests rehearse with gloo)")
    ap.add_argument("--launcher-only", action="store_true", help="(tests) start the ranks, let them report, do no GPU work")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`bench.py --gpus N` run plainly: start N ranks of this script (one process per GPU) and return
    their exit status.  Nothing in this (parent) process has initialised a GPU; the ranks are fresh
    children, rank 0 prints the JSON line, and a rank that fails to join fails the whole run."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def munk_tables(r_max, nr=100, sofar_slope=0.0):
    import pygenray_amd as pr
    z = np.arange(0, 6000, 1.0)
    r = np.linspace(0.0, r_max, nr)
    # (configs[2]: c[i, :] = munk_ssp(z, sofar_depth = 1300 + slope * r_i), the pattern of REF/tests/test_physics.py:497)
    c2 = np.array([pr.munk_ssp(z, 1300.0 + sofar_slope * ri) for ri in r]) if sofar_slope else np.tile(pr.munk_ssp(z), (nr, 1))
    ssp = pr.DataArray(c2, dims=["range", "depth"],
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


def cpu_baseline_c(arrs, n_rays=25000):
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
                      f"1000 km, {steps} ray-steps in {dt:.1f} s, oracle/ray_oracle.c (libm mode) with OpenMP, "
                      f"{cores} threads (cgroup CPU quota of the box)"}


_POOL_ARRS = None


def _pool_init(arrs):
    """Pool initializer: the tables reach every worker ONCE (not pickled with each task inside the timed map)."""
    global _POOL_ARRS
    _POOL_ARRS = arrs
    import oracle.scipy_port  # noqa: F401  (imports done before the clock starts)


def _scipy_chunk(y0):
    """One pool task: a chunk of rays through the reference's solve_ivp call pattern."""
    from oracle import scipy_port
    out = scipy_port.shoot_fan(*_POOL_ARRS, y0, 0.0, RANGE_M, S_SAVE)
    return int(out["n_steps"].sum()), int((out["status"] == 0).sum())


def cpu_baseline_scipy(arrs, seconds_budget=25.0):
    """the NumPy/SciPy call pattern of the reference -- solve_ivp(RK45, rtol 1e-9, atol
    1e-6, dense output, 4 terminal events, REF/launch_rays.py:670-679) restarted at every bounce, NumPy
    right-hand side -- on ALL usable host cores through a spawn pool, one chunk of rays per task (the
    reference's pool maps one ray per task, REF/launch_rays.py:157-164), on a strided subset of the same
    1e5-ray fan sized to the time budget.  UN-JITTED: the reference compiles its right-hand side and event
    functions with numba (REF/integration_processes.py:26), which this image does not have, so this
    under-states a real pygenray install by the RHS share of a step (SciPy's per-step Python overhead,
    ~100 us, remains either way); `cpu_baseline_c` is the compiled restatement of the same path."""
    import multiprocessing as mp
    from pygenray_amd.device_fan import fan_y0
    cores = host_cores()
    # ~0.12 s per 1000-km ray and core (11 k ray-steps/s): size the sample to ~seconds_budget of wall clock
    n_rays = int(max(cores * 4, min(4000, cores * seconds_budget / 0.14)))
    stride = max(1, RAYS_PER_GPU // n_rays)
    theta = np.linspace(-20, 20, RAYS_PER_GPU)[::stride][:n_rays]
    y0 = fan_y0(arrs, SOURCE_DEPTH, 0.0, -theta)
    order = np.random.default_rng(0).permutation(len(y0))   # mix steep and shallow rays over the tasks
    chunks = [y0[order[k::cores * 4]] for k in range(cores * 4)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores, initializer=_pool_init, initargs=(arrs,)) as pool:
        pool.map(_noop, range(cores))          # workers up, tables and imports in place, before the clock starts
        t0 = time.time()
        res = pool.map(_scipy_chunk, chunks, chunksize=1)
        dt = time.time() - t0
    steps = sum(r[0] for r in res)
    return {"value": steps / dt, "unit": "ray-steps/s", "cores": cores, "kind": "port", "jit": False,
            "sample": f"{len(y0)} rays (every {stride}th of the 1e5-ray fan), 1000 km, {steps} ray-steps in {dt:.1f} s: "
                      f"oracle/scipy_port.py = scipy.integrate.solve_ivp(RK45, 4 terminal events, dense output) with a "
                      f"NumPy right-hand side (pygenray's call pattern; un-jitted -- the reference jits its RHS with numba, "
                      f"absent here -- so a real install is faster by the RHS share of a step), spawn pool of {cores} processes"}


def _noop(_):
    return 0


def kernel_leg(arrs, n_rays, save, passes=6, amin=-20.0, amax=20.0, rays_lo=None, blocked=False):
    """One more workload through the same device entry, a few passes: kernel time from HIP events on the launch
    stream, accepted steps, the contract's algorithmic bytes and the fraction of the HBM peak they amount to."""
    import torch
    from pygenray_amd import _lib
    from pygenray_amd.device_fan import DeviceFan, fan_y0
    theta = np.linspace(amin, amax, n_rays)
    if rays_lo is not None:
        theta = theta[rays_lo[0]:rays_lo[1]]
    env = _lib.EnvHandle(*arrs)
    y0 = fan_y0(arrs, SOURCE_DEPTH, 0.0, -theta)
    fan = DeviceFan(env, y0, 0.0, RANGE_M, S_SAVE, save=save, sample_major=True, sample_blocked=blocked)
    for _ in range(3):     # (the clock has dropped while the host built this leg's tables: let it come back)
        fan.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(passes):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fan.run(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.mean(ts))
    steps = fan.ray_steps()
    alive = max(fan.N - int((fan.status != 0).sum().item()), 1)
    b_alg = 80.0 + (24.0 * S_SAVE / (steps / alive) if save else 0.0)
    out = {"kernel_ms": ms, "kernel_ms_min": float(np.min(ts)), "frac": steps * b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ray_steps": steps,
           "ray_steps_per_s": steps / (ms * 1e-3), "bytes_per_ray_step": b_alg, "passes": passes}
    del fan
    env.close()
    return out


def extra_legs():
    """The kernels users hit beside the headline instance, timed by the same run: the
    reference's DEFAULT environment handling (flat_earth_transform=True: smoothly non-uniform zin -> the
    cubic-index depth look-up, kernel ZM = 5), configs[2] (range-dependent tables in HBM / L2), the per-GPU
    fan size of configs[3] / configs[4] (1e6 rays, end state only) and the lone wave of the fan's 64 steepest
    rays, whose latency is the 1e5-ray fan's floor."""
    import pygenray_amd as pr
    from pygenray_amd.environment import _unpack_envi
    legs = {}
    env_fe, _ = munk_tables(RANGE_M)
    env_fe.flat_earth_transform(lat=35)
    arrs_fe = _unpack_envi(env_fe, flatearth=True)
    legs["flatearth_default"] = {
        "workload": "configs[1] tables after OceanEnvironment2D's default flat-earth transform (non-uniform zin), 1e5 rays, 1000 km",
        "kernel": "pgr_fan_kernel<true, 5, SAVE> (table + zin in LDS, cubic index estimate)",
        "end_state": kernel_leg(arrs_fe, RAYS_PER_GPU, False), "trajectories": kernel_leg(arrs_fe, RAYS_PER_GPU, True)}
    _, arrs_rd = munk_tables(RANGE_M, nr=101, sofar_slope=2e-4)
    legs["range_dependent"] = {
        "workload": "configs[2]: sofar axis + 2e-4 r over 101 columns, 1e5 rays, 1000 km",
        "kernel": "pgr_fan_kernel<false, 4, SAVE> (tables in HBM / L2); trajectories: SAVE = 3, the sample-blocked layout "
                  "[S/4][N][4] (PGR_SAMPLE_BLOCKED: each lane stores four consecutive samples as one 32-byte piece per array); "
                  "trajectories_row_layout: SAVE = 1, plain [S][N] rows",
        "end_state": kernel_leg(arrs_rd, RAYS_PER_GPU, False), "trajectories": kernel_leg(arrs_rd, RAYS_PER_GPU, True, blocked=True),
        "trajectories_row_layout": kernel_leg(arrs_rd, RAYS_PER_GPU, True)}
    _, arrs1 = munk_tables(RANGE_M)
    legs["rays_1e6"] = {
        "workload": "configs[1] tables, 1e6 launch angles (the per-GPU fan of configs[3] / configs[4]): end state only, and with S = 1001 trajectories",
        "kernel": "pgr_fan_kernel<true, 4, 0, true> (persistent waves: one workgroup per CU, 64-ray packets claimed from the cost-sorted list)",
        "end_state": kernel_leg(arrs1, 1_000_000, False, passes=3),
        # the headline's own kernel shape (S = 1001 trajectories, 24 GB of samples in HBM) at ten times its rays
        "trajectories": kernel_leg(arrs1, 1_000_000, True, passes=3)}
    # the headline fan itself WITHOUT trajectories (end state only, B_alg = 80 B): the instance eigenray searches and
    # histogram fans of this size run, and the number every end-state comparison of DESIGN.md refers to
    legs["headline_end_state"] = dict(kernel_leg(arrs1, RAYS_PER_GPU, False),
                                      workload="configs[1], 1e5 rays, 1000 km, end state only", kernel="pgr_fan_kernel<true, 4, 0, false>")
    lone = {"end_state": kernel_leg(arrs1, RAYS_PER_GPU, False, rays_lo=(0, 64)),
            "trajectories": kernel_leg(arrs1, RAYS_PER_GPU, True, rays_lo=(0, 64))}
    return legs, lone


def config4_leg(env, arrs, rank, world, fence, passes=3, rays_per_gpu=1_000_000, bins=4096):
    """BASELINE configs[4] as a leg of every N > 1 run (all ranks take part): 1e6 launch angles per GPU of the global
    fan linspace(-20, 20, N x 1e6), dealt strided; end state only, the kernel writes the 40-byte end records itself;
    each pass ends with the 4096-bin arrival-time histogram (HIP kernel + all-reduce of the counts) and the all-gather
    of the end records, the gather of pass k travelling while pass k + 1 integrates.  Wall time = max over ranks
    (barrier + synchronize on both sides).  Returns the leg on every rank (rank 0 prints it)."""
    import torch
    import torch.distributed as dist
    from pygenray_amd.device_fan import DeviceFan, fan_y0
    from pygenray_amd.distributed import shard_indices, start_all_gather_records, arrival_time_histogram
    n_global = rays_per_gpu * world
    theta = np.linspace(-20, 20, n_global)
    idx = shard_indices(n_global, rank, world)
    y0 = fan_y0(arrs, SOURCE_DEPTH, 0.0, -theta[idx])
    n_pad = (n_global + world - 1) // world
    fan = DeviceFan(env, y0, 0.0, RANGE_M, 1, save=False, packed_end=True, n_pad=n_pad)
    t_lo, t_hi = RANGE_M / 1560.0, RANGE_M / 1400.0
    hist, gathered = None, None

    kev = []

    def one_pass(pending):
        nonlocal hist, gathered
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fan.run(); e1.record()
        kev.append((e0, e1))
        hist = arrival_time_histogram(fan.records[:fan.N, 0], fan.status, bins, t_lo, t_hi, reduce=True)
        started = start_all_gather_records(fan.records, n_global)
        if pending is not None:
            gathered = pending.finish()
        return started

    pending = one_pass(None)
    pending.finish()
    fence()
    kev.clear()
    t0 = time.perf_counter()
    pending = None
    for _ in range(passes):
        pending = one_pass(pending)
    gathered = pending.finish()
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ss = torch.tensor([fan.ray_steps(), int((fan.status != 0).sum().item())], dtype=torch.int64, device="cuda")
    dist.all_reduce(ss, op=dist.ReduceOp.SUM)
    end_all, _, _, st_all = gathered
    ok_all = int((st_all == 0).sum().item())
    # per GPU: the fan kernel's own time (HIP events on the launch stream, this rank's passes) and the fraction of the HBM
    # peak its algorithmic bytes (80 B per ray-step, end state only) amount to -- gathered, so that the line is gradeable
    # per GPU whatever the collectives cost
    kms = float(np.mean([a.elapsed_time(b) for a, b in kev]))
    mine = torch.tensor([kms, float(fan.ray_steps())], dtype=torch.float64, device="cuda")
    per = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(per, mine)
    per_gpu = [{"rank": k, "kernel_ms": float(p_[0].item()), "ray_steps": int(p_[1].item()),
                "frac": float(p_[1].item()) * 80.0 / (float(p_[0].item()) * 1e-3) / 1e9 / HBM_PEAK_GBS} for k, p_ in enumerate(per)]
    return {"roofline_per_gpu": {"bound": "hbm", "bytes_per_ray_step": 80.0, "peak_gbs": HBM_PEAK_GBS,
                                 "kernel": "pgr_fan_kernel<true, 4, 0, true> (persistent waves)", "ranks": per_gpu,
                                 "frac_min": min(q["frac"] for q in per_gpu), "kernel_ms_max": max(q["kernel_ms"] for q in per_gpu)},"workload": f"configs[4]: {n_global} launch angles linspace(-20, 20) over {world} ranks ({rays_per_gpu} per GPU, strided), "
                        "1000 km, end state only; per pass: fan kernel (40-byte end records written in place), 4096-bin arrival-time "
                        "histogram + all-reduce, all-gather of the end records (overlapping the next pass)",
            "ms_per_pass": float(tt.item()) / passes * 1e3, "passes": passes,
            "ray_steps_per_pass": int(ss[0].item()), "ray_steps_per_s": int(ss[0].item()) * passes / float(tt.item()),
            "dropped_rays": int(ss[1].item()), "gathered_rays": int(end_all.shape[0]), "gathered_ok": ok_all,
            "histogram_counted_rays": int(hist.sum().item()), "histogram_bins": bins,
            "all_gather_bytes_per_rank": n_pad * 40, "ranks": world}


def api_leg(env_obj, calls=3, what="configs[1]"):
    """What a `pr.shoot_rays` caller sees on configs[1] / configs[2] (1e5 launch angles, 1000 km, S = 1001): wall clock of the drop-in
    call itself -- environment unpack, initial states, kernel, compaction of dropped rays, PCIe, RayFan -- eager
    (`device_resident=False`: all three (M, S) arrays on the host when the call returns) and device resident (the
    default for a fan this size: per-ray arrays and end states on the host, trajectories fetched when first read),
    the best of the calls after the first (tables resident, buffers pooled).  This is the path that replaces the
    reference's process pool + pickled Rays (REF/launch_rays.py:157-186)."""
    import pygenray_amd as pr
    angles = np.linspace(-20, 20, RAYS_PER_GPU)
    kw = dict(debug=False, flatearth=False)
    pr.shoot_rays(SOURCE_DEPTH, 0.0, angles[:1000], RANGE_M, S_SAVE, env_obj, **kw)      # table upload, library warm-up
    out = {"workload": f"pr.shoot_rays(1000 m, 0, linspace(-20, 20, 100000), 1000 km, 1001, env, flatearth=False): {what} through the drop-in API",
           "calls": calls}
    for name, mode in (("eager", False), ("device_resident", True)):
        walls, reads = [], []
        for _ in range(calls):
            t0 = time.perf_counter()
            fan = pr.shoot_rays(SOURCE_DEPTH, 0.0, angles, RANGE_M, S_SAVE, env_obj, device_resident=mode, **kw)
            t1 = time.perf_counter()
            nbytes = sum(a.nbytes for a in (fan.ts, fan.zs, fan.ps))       # (device resident: the three fetches happen here)
            t2 = time.perf_counter()
            walls.append(t1 - t0); reads.append(t2 - t1)
            kept = len(fan)
            del fan
        out[name] = {"wall_ms": 1e3 * min(walls[1:]), "first_call_ms": 1e3 * walls[0],
                     "then_reading_ts_zs_ps_ms": 1e3 * min(reads[1:]), "rays_kept": kept,
                     "trajectory_bytes_to_host": int(nbytes),
                     "bytes_to_host_inside_the_call": int(nbytes if not mode else kept * (3 * 8 + 8 + 4 + 4)),
                     "pcie_floor_ms_at_56GBs": 1e3 * nbytes / 56e9}
    return out


def fma_leg():
    """OPTIONAL, separately labelled, never the headline: the same sources built with FMA contraction allowed
    (pygenray_amd/csrc/libpgr_hip_fma.so, built by __graft_entry__.build beside the product) on the headline fan, in a
    child process (one library per process), with its deviation from the reference's vectors g11-g13 against the
    reference's own self-noise -- what the bit-identical default costs and what it buys (scripts/fma_leg.py)."""
    lib = os.path.join(ROOT, "pygenray_amd", "csrc", "libpgr_hip_fma.so")
    if not os.path.exists(lib):
        return None
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fma_leg.py"), "--lib", lib], capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(line[-1]) if (r.returncode == 0 and line) else {"error": (r.stderr or r.stdout)[-400:]}
    except Exception as exc:   # noqa: BLE001  (an optional leg must not cost the line)
        return {"error": f"{type(exc).__name__}: {exc}"}


def eigenray_leg(env_obj, n_rays):
    """BASELINE configs[3]: fixed source / receiver, a fan of n_rays launch angles (end state only)
    then pygenray's regula falsi on every bracket (REF/eigenrays.py:62-203), receiver depth 1000 m,
    ztol 1 m, max_iter 20: wall clock through the drop-in API, second run (tables resident)."""
    import pygenray_amd as pr
    from pygenray_amd import eigenrays as er_mod
    angles = np.linspace(-20, 20, n_rays)
    out = None
    for _ in range(2):
        er_mod.LAST_SEARCH_STATS.clear()
        t0 = time.perf_counter()
        fan = pr.shoot_rays(SOURCE_DEPTH, 0.0, angles, RANGE_M, 2, env_obj, debug=False, flatearth=False)
        t1 = time.perf_counter()
        er = pr.find_eigenrays(fan, [1000.0], SOURCE_DEPTH, 0.0, RANGE_M, 2, env_obj, ztol=1, max_iter=20,
                               debug=False, flatearth=False, quiet=True)
        t2 = time.perf_counter()
        out = {"wall_s": t2 - t0, "fan_s": t1 - t0, "search_s": t2 - t1, "fan_rays": n_rays,
               "brackets": int(er.num_eigenrays[1000.0]), "found": int(er.num_eigenrays_found[0]),
               "failed": len(er.failed_eray_theta_brackets[0]),
               "launches": int(er_mod.LAST_SEARCH_STATS.get("launches", 0)),
               "config": "configs[3]: Munk dz=1 m, source (0, 1000 m), receiver (1000 km, 1000 m), "
                         "ztol 1 m, max_iter 20; through pr.shoot_rays + pr.find_eigenrays (device-resident "
                         "false-position loop, pgr_eigen_refine), host buffers included"}
    # the same fan searched for FOUR receiver depths in one call: their brackets iterate together on the device
    # (pgr_eigen_refine_depths), so the search costs what its longest single-depth search costs
    depths = [500.0, 1000.0, 1500.0, 2000.0]
    er_mod.LAST_SEARCH_STATS.clear()
    t0 = time.perf_counter()
    er4 = pr.find_eigenrays(fan, depths, SOURCE_DEPTH, 0.0, RANGE_M, 2, env_obj, ztol=1, max_iter=20,
                            debug=False, flatearth=False, quiet=True)
    out["four_receiver_depths"] = {"search_s": time.perf_counter() - t0, "receiver_depths_m": depths,
                                   "brackets": int(sum(er4.num_eigenrays[d] for d in depths)),
                                   "found": int(sum(er4.num_eigenrays_found[k] for k in range(len(depths)))),
                                   "launches": int(er_mod.LAST_SEARCH_STATS.get("launches", 0))}
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PGR_BENCH_ONE_GPU=1 (rehearsal on a one-GPU box, with --backend gloo): every rank drives device 0.  The N > 1 code path
    # runs for real -- strided shards, packed end records, all-gather, sharded eigenray search -- but the ranks share one
    # GPU, so the line it prints is a functional check, never a scaling number (it says so in `config.sharding`)
    one_gpu = os.environ.get("PGR_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # launched under an external torchrun with a different rank count: the JSON line must not lie
        if rank == 0:
            print(f"error: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(2)

    # stdout carries exactly ONE line, rank 0's JSON: whatever the libraries under us print there (RCCL's
    # version banner at the first collective, for one) goes to stderr instead -- file descriptor 1 is pointed at
    # stderr for the whole run and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist
    if args.launcher_only:
        # (CPU rehearsal of the launcher, tests/test_host.py: every rank joins, rank 0 reports)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            emit({"launcher_only": True, "n_gpus": world, "ranks_joined": int(t.item())})
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if int(t.item()) == args.gpus else 3)

    from pygenray_amd import _lib
    from pygenray_amd.device_fan import DeviceFan, fan_y0
    from pygenray_amd.distributed import shard_indices, start_all_gather_records, arrival_time_histogram

    torch.cuda.set_device(dev_index)
    # PGR_BENCH_FORCE_DIST=1 rehearses the N > 1 code path (RCCL init, all-gather, reductions)
    # with a single rank on a one-GPU box
    use_dist = world > 1 or os.environ.get("PGR_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "RANK" not in os.environ:   # the one-rank rehearsal outside a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if one_gpu and world > 1 and args.backend == "nccl":
            args.backend = "gloo"    # RCCL refuses two ranks on one device ("Duplicate GPU detected"): the rehearsal rides gloo
        if args.backend == "nccl":
            dist.init_process_group(args.backend, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
        joined = torch.ones(1, device="cuda")
        dist.all_reduce(joined)
        ranks_joined = int(joined.item())
        if ranks_joined != world:
            sys.exit(3)

    _lib.load()
    env_obj, arrs = munk_tables(RANGE_M, nr=101, sofar_slope=2e-4) if args.range_dependent else munk_tables(RANGE_M)
    if args.flat_earth:
        from pygenray_amd.environment import _unpack_envi
        env_obj.flat_earth_transform(lat=35)
        arrs = _unpack_envi(env_obj, flatearth=True)
    env = _lib.EnvHandle(*arrs, device=dev_index)
    if args.waves_per_block:
        env.set_option("waves_per_block", args.waves_per_block)
    n_global = args.rays * world
    theta = np.linspace(-20, 20, n_global)
    idx = shard_indices(n_global, rank, world)
    y0 = fan_y0(arrs, SOURCE_DEPTH, 0.0, -theta[idx])  # ODE angle = -user (>= 70-ray branch)
    save = not args.no_save
    # N > 1: the kernel writes the 40-byte end records of the all-gather itself (PGR_PACKED_END)
    fan = DeviceFan(env, y0, 0.0, RANGE_M, S_SAVE, save=save, sample_major=(args.layout == "sample"),
                    packed_end=use_dist, n_pad=(n_global + world - 1) // world, sample_blocked=(args.blocked and save))
    HIST_BINS, T_LO, T_HI = 4096, RANGE_M / 1560.0, RANGE_M / 1400.0
    hist = None

    def step_collectives(pending):
        """what follows a fan pass: the histogram of configs[4] and the all-gather of the end records"""
        nonlocal hist
        if args.histogram:
            t_end = fan.records[:fan.N, 0] if use_dist else fan.end[:, 0]
            hist = arrival_time_histogram(t_end, fan.status, HIST_BINS, T_LO, T_HI, reduce=use_dist)
        if use_dist:
            # the end records of pass k travel (RCCL stream) while pass k+1 integrates; every
            # gathered fan is reassembled in launch-angle order before the clock stops
            started = start_all_gather_records(fan.records, n_global)
            if pending is not None:
                pending.finish()
            return started
        return None

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    pending = None
    for _ in range(args.warmup):
        fan.run()
        pending = step_collectives(pending)
    if pending is not None:
        pending.finish()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    pending = None
    for k in range(args.steps):
        ev[k][0].record()
        fan.run()
        ev[k][1].record()
        pending = step_collectives(pending)
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

    # configs[3] over the ranks (every rank takes part): the fan's launch angles are dealt to the ranks, ONE all-gather
    # of the end records puts the fan on every rank, every rank brackets on the whole fan, the brackets are dealt to the
    # ranks for the device-resident false-position loop, one small gather collects the eigenrays
    eig_sharded = None
    if use_dist and not args.no_eigenray:
        from pygenray_amd.distributed import shoot_rays_sharded, find_eigenrays_sharded
        angles = np.linspace(-20, 20, args.eigen_rays)
        for _ in range(2):   # second run: tables resident, buffers allocated
            fence()
            t_a = time.perf_counter()
            gfan = shoot_rays_sharded(SOURCE_DEPTH, 0.0, angles, RANGE_M, env_obj, flatearth=False, device=dev_index)
            t_b = time.perf_counter()
            ger = find_eigenrays_sharded(gfan, [1000.0], SOURCE_DEPTH, 0.0, RANGE_M, 2, env_obj, ztol=1, max_iter=20,
                                         debug=False, flatearth=False, quiet=True, device=dev_index)
            fence()
            t_c = time.perf_counter()
        tt = torch.tensor([t_c - t_a, t_b - t_a], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        # per GPU: this rank's share of the fan (rays, accepted steps are not returned by the end-record path: the share of the
        # 1e6-angle fan is N / world rays) against the wall time of the sharded fan call (kernel + all-gather + host)
        share = len(shard_indices(args.eigen_rays, rank, world))
        eig_sharded = {"wall_s": float(tt[0].item()), "fan_s": float(tt[1].item()), "fan_rays": int(args.eigen_rays),
                       "fan_rays_per_gpu": int(share), "fan_rays_per_s_per_gpu": share / float(tt[1].item()),
                       "ranks": world, "brackets": int(ger.num_eigenrays[1000.0]), "found": int(ger.num_eigenrays_found[0]),
                       "failed": len(ger.failed_eray_theta_brackets[0]),
                       "config": "configs[3] sharded: pygenray_amd.distributed.shoot_rays_sharded (strided shards, all-gather "
                                 "of 40-byte end records) + find_eigenrays_sharded (brackets dealt to the ranks)"}

    # configs[4] (1e6 rays per GPU, end records, histogram all-reduce, all-gather): a leg of every N > 1 run
    cfg4 = None
    if use_dist and world > 1 and not args.no_legs:
        cfg4 = config4_leg(env, arrs, rank, world, fence)

    if rank == 0:
        value = total_steps * args.steps / dt
        # B_alg = 80 B (state in + out) + 24 B per saved (T,z,p) sample
        mean_steps = local_steps / max(fan.N - int((fan.status != 0).sum().item()), 1)
        b_alg = 80.0 + (24.0 * S_SAVE / mean_steps if save else 0.0)
        achieved = local_steps * b_alg / (kern_ms * 1e-3) / 1e9
        # HBM bytes and VALU wave-instructions per launch from the committed rocprofv3 PMC passes of this same
        # command (profiles/rNN_traffic.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE; SQ_INSTS_VALU).
        # Counters describe a BINARY: the file records the sha256 of the gfx950 machine code they were taken with
        # (pygenray_amd._lib.device_code_sha256) and they are reported only when the loaded library carries the
        # same code and the run has the same shape -- otherwise `traffic` is null and says why.
        traffic = traffic_gb = valu = None
        tnote = "no committed PMC pass for this configuration"
        code_sha = _lib.device_code_sha256()
        tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
        if tfiles:
            tpath = os.path.join(ROOT, "profiles", tfiles[-1])
            try:
                tj = json.load(open(tpath))
                key = (("rangedep-" if args.range_dependent else "") + ("flatearth-" if args.flat_earth else "")
                       + ("blocked" if (args.blocked and save) else args.layout) + ("" if save else "-nosave")
                       + ("" if fan.N == RAYS_PER_GPU else f"@{fan.N}"))
                if key not in tj or tj[key].get("rays") != fan.N:
                    tnote = f"profiles/{tfiles[-1]} holds no PMC pass of this workload"
                elif tj.get("device_code_sha256") != code_sha:
                    tnote = (f"profiles/{tfiles[-1]} was taken with other kernel code (device_code_sha256 "
                             f"{str(tj.get('device_code_sha256'))[:12]}..., loaded {code_sha[:12]}...): counters withheld")
                else:
                    traffic_gb = tj[key]["hbm_gb_per_launch"]
                    traffic = traffic_gb / (kern_ms * 1e-3)
                    valu = tj[key].get("valu_wave_instructions_per_launch")
                    tnote = tj[key].get("note", f"profiles/{tfiles[-1]}")
            except Exception as exc:
                traffic = traffic_gb = valu = None
                tnote = f"profiles/{tfiles[-1]} unreadable ({type(exc).__name__})"
        what = ("configs[2] range-dependent Munk fan" if args.range_dependent else "Munk fan") + (" on the flat-earth grid" if args.flat_earth else "")
        metric = (BASELINE_METRIC if (fan.N == RAYS_PER_GPU and not args.range_dependent and not args.histogram and not args.flat_earth)
                  else f"ray-steps/sec (whole node), {fan.N}-ray {what} to 1000 km"
                       + (" + arrival-time histogram" if args.histogram else ""))
        out = {
            "metric": metric,
            "value": value, "unit": "ray-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"configs[2]: range-dependent Munk SSP (sofar axis + 2e-4 r, 101 columns) dz=1 m, "
                                    if args.range_dependent else "configs[1]: Munk SSP dz=1 m, ")
                                   + ("flat-earth transformed tables (non-uniform zin), " if args.flat_earth else "")
                                   + f"{fan.N} launch angles per GPU "
                                   "linspace(-20,20), 1000 km, rtol 1e-9, fp64"
                                   + (", + 4096-bin arrival-time histogram (configs[4] shape)" if args.histogram else ""),
                       "rays_per_gpu": fan.N, "num_range_save": S_SAVE if save else 0,
                       "trajectory_layout": ("sample-blocked [S/4][N][4]" if args.blocked else args.layout) if save else "none",
                       "ray_steps_per_pass": total_steps, "dropped_rays": n_drop,
                       "histogram_bins": HIST_BINS if args.histogram else 0,
                       "sharding": ("strided launch angles, all-gather of end records" + (" -- REHEARSAL: all ranks share ONE GPU "
                                    f"(PGR_BENCH_ONE_GPU, backend {args.backend}); not a scaling number" if one_gpu else ""))
                                   if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_gb_per_launch": traffic_gb, "traffic_source": tnote,
                         "algorithmic_gb_per_launch": local_steps * b_alg / 1e9,
                         "kernel": f"pgr_fan_kernel<{'false' if args.range_dependent else 'true'}, {5 if args.flat_earth else 4}, {(3 if args.blocked else 1) if save else 0}, "
                                   f"{'true' if (fan.N + 63) // 64 > 8 * 256 else 'false'}> "
                                   f"(table in {'HBM/L2' if args.range_dependent else 'LDS'}, "
                                   f"{'non-uniform zin, cubic index estimate' if args.flat_earth else 'zin = j * 1 m'}, "
                                   f"{'linspace save grid' if save else 'end state only'})", "kernel_ms": kern_ms,
                         # every timed step's own kernel time (HIP events on the launch stream): the mean above is over exactly these
                         "kernel_ms_each": [round(a.elapsed_time(b), 4) for a, b in ev],
                         "bytes_per_ray_step": b_alg,
                         "note": "algorithmic bytes per SURVEY 8(d); the stepper keeps state in "
                                 "VGPRs and the SSP table in LDS, so it is fp64-VALU bound, not HBM bound"},
            "build": _lib.build_info(), "device_code_sha256": code_sha,
        }
        if valu:
            slots = SIMDS * CLOCK_HZ / FP64_CYCLES_PER_WAVE_INSTR * kern_ms * 1e-3
            out["roofline_valu"] = {"bound": "fp64 VALU issue", "achieved": valu, "peak": slots,
                                    "unit": "wave-instructions per launch", "frac": valu / slots,
                                    "note": "SQ_INSTS_VALU of the fan kernel (committed PMC pass) over 1024 SIMDs x 2.4 GHz / 4 "
                                            "cycles x this run's kernel time: the bound that really holds (every VALU "
                                            "instruction of the stepper is fp64 or issues at the same 4-cycle cadence)"}
        if args.histogram and hist is not None:
            out["histogram"] = {"bins": HIST_BINS, "range_s": [T_LO, T_HI], "counted_rays": int(hist.sum().item())}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_scipy(arrs)
            out["cpu_baseline_c"] = cpu_baseline_c(arrs)
        if world == 1 and not args.no_eigenray:
            out["eigenray"] = eigenray_leg(env_obj, args.eigen_rays)
        if eig_sharded is not None:
            out["eigenray_sharded"] = eig_sharded
        if use_dist:
            out["ranks_joined"] = ranks_joined      # (from the all-reduce every rank took part in at start-up)
        if cfg4 is not None:
            out.setdefault("legs", {})["config4"] = cfg4
        if world == 1 and not args.no_legs:
            del fan   # (2.4 GB of trajectories back before the legs allocate theirs)
            legs, lone = extra_legs()
            legs["api"] = api_leg(env_obj)
            # configs[2] through the same API: the tables stay in HBM / L2, so the fan runs the sample-blocked kernel
            # <false, 4, 3> and is un-blocked by the pass that squeezes its dropped rays out (PGR_OPT_API_BLOCKED)
            env_rd, _ = munk_tables(RANGE_M, nr=101, sofar_slope=2e-4)
            legs["api_config2"] = api_leg(env_rd, what="configs[2] (range-dependent tables: sample-blocked kernel, un-blocked on the way out)")
            fl = fma_leg()
            if fl is not None:
                legs["fma_contracted"] = fl
            out["legs"] = legs
            # the fan cannot finish before its steepest rays do: the first wave of the fan (64 steepest rays) ALONE
            out["lone_wave_ms"] = {"end_state": lone["end_state"]["kernel_ms"], "trajectories": lone["trajectories"]["kernel_ms"],
                                   "note": "rays 0..63 of the 1e5-ray fan alone on the chip: the floor of the fan's kernel time"}
            out["roofline"]["frac_rays_1e6_end_state"] = legs["rays_1e6"]["end_state"]["frac"]
            out["roofline"]["frac_rays_1e6_trajectories"] = legs["rays_1e6"]["trajectories"]["frac"]
        emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
