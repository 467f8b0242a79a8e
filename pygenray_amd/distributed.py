"""Multi-GPU fan: launch angles shard across ranks, one all-gather reassembles the end states.

Rays are independent (the reference maps one ray per pool task, REF/launch_rays.py:157-164);
the only cross-ray step is eigenray bracketing on *adjacent* launch angles
(REF/eigenrays.py:65-72).  So: one process per GPU, every rank holds the full (small) tables
and integrates a strided subset of the sorted launch angles (steep, bouncing rays cost ~1.4x
more steps and sit at the fan edges -- a strided deal balances the ranks), then ONE
all-gather (RCCL over xGMI with backend "nccl", gloo on CPU) of the 40-byte per-ray end record
``(T_end, z_end, p_end, n_bott, n_surf, status)`` puts the whole fan, in launch-angle order,
on every rank.  Trajectories stay sharded.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n_rays, rank, world_size):
    """Global ray indices integrated by `rank` (strided deal)."""
    return np.arange(rank, n_rays, world_size)


def pack_end_records(end, n_bott, n_surf, status, n_pad):
    """-> float64 [n_pad, 5]: cols 0..2 = T, z, p; cols 3..4 carry 4 int32 (n_bott, n_surf,
    status, valid)."""
    n = end.shape[0]
    buf = torch.zeros((n_pad, 5), dtype=torch.float64, device=end.device)
    buf[:n, 0:3] = end
    ints = buf[:, 3:5].view(torch.int32)
    ints[:n, 0] = n_bott.to(torch.int32)
    ints[:n, 1] = n_surf.to(torch.int32)
    ints[:n, 2] = status.to(torch.int32)
    ints[:n, 3] = 1
    return buf


class FanGather:
    """An all-gather of end records in flight (``start_all_gather_fan``): ``finish()`` makes the
    current stream wait for it and returns the fan in global launch-angle order."""

    def __init__(self, work, gathered, world, n_pad, n_rays, keep):
        self.work, self.gathered, self.world, self.n_pad, self.n_rays = work, gathered, world, n_pad, n_rays
        self._keep = keep  # the packed send buffer must outlive the collective

    def finish(self):
        if self.work is not None:
            self.work.wait()   # stream-ordered for RCCL: the host does not block
            self.work = None
        # rank r holds global rays r, r+W, r+2W, ... -> interleave back
        full = self.gathered.permute(1, 0, 2).reshape(self.world * self.n_pad, 5)[:self.n_rays]
        ints = full[:, 3:5].contiguous().view(torch.int32)
        return (full[:, 0:3].contiguous(), ints[:, 0].contiguous(), ints[:, 1].contiguous(),
                ints[:, 2].contiguous())


def start_all_gather_records(records, n_rays, group=None):
    """The same for end records the kernel already packed (``DeviceFan(packed_end=True,
    n_pad=ceil(n_rays / world))``): the send buffer is cloned (one copy kernel) so that the next
    fan may overwrite ``records`` while they travel."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_pad = (n_rays + world - 1) // world
    assert records.shape == (n_pad, 5), (tuple(records.shape), n_pad)
    local = records.clone()
    if dist.is_initialized():
        flat = torch.empty((world * n_pad, 5), dtype=torch.float64, device=local.device)
        work = dist.all_gather_into_tensor(flat, local, group=group, async_op=True)
        return FanGather(work, flat.view(world, n_pad, 5), world, n_pad, n_rays, local)
    return FanGather(None, local.unsqueeze(0), world, n_pad, n_rays, local)


def start_all_gather_fan(end, n_bott, n_surf, status, n_rays, group=None):
    """Pack the local end records and START their all-gather (asynchronous: the collective runs
    on RCCL's stream behind the work already queued, so the next fan can be launched while the
    records travel).  The inputs may be overwritten as soon as this returns (they are copied)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_pad = (n_rays + world - 1) // world
    local = pack_end_records(end, n_bott, n_surf, status, n_pad)
    if dist.is_initialized():  # also with a single rank: the collective is then a copy
        flat = torch.empty((world * n_pad, 5), dtype=torch.float64, device=local.device)
        work = dist.all_gather_into_tensor(flat, local, group=group, async_op=True)
        return FanGather(work, flat.view(world, n_pad, 5), world, n_pad, n_rays, local)
    return FanGather(None, local.unsqueeze(0), world, n_pad, n_rays, local)


def all_gather_fan(end, n_bott, n_surf, status, n_rays, group=None):
    """All-gather the local end records and return them in global launch-angle order:
    ``(end [N,3] float64, n_bott [N], n_surf [N], status [N])`` on every rank."""
    return start_all_gather_fan(end, n_bott, n_surf, status, n_rays, group=group).finish()


def shoot_fan_sharded(compute, y0_all, group=None):
    """Integrate this rank's strided shard with `compute(y0_local) -> (end, n_bott, n_surf,
    status)` (torch tensors) and all-gather.  `compute` is the HIP fan in production
    (``hip_compute``); tests inject a CPU function to exercise the collective under gloo."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    y0_all = np.asarray(y0_all, dtype=np.float64).reshape(-1, 3)
    n = len(y0_all)
    idx = shard_indices(n, rank, world)
    end, nb, ns, st = compute(y0_all[idx])
    return all_gather_fan(end, nb, ns, st, n, group=group)


def hip_compute(env_handle, source_range, receiver_range, rtol=1e-9, atol=1e-6,
                terminate_backwards=True, max_steps=1_000_000):
    """End-state-only HIP fan as a `compute` callback for ``shoot_fan_sharded``."""
    from .device_fan import cached_end_state_fan

    def compute(y0_local):
        # (one set of device buffers per environment handle, grow-only: no allocation per fan)
        fan = cached_end_state_fan(env_handle, y0_local, source_range, receiver_range, rtol=rtol, atol=atol,
                                   terminate_backwards=terminate_backwards, max_steps=max_steps)
        fan.run()
        return fan.end, fan.n_bott, fan.n_surf, fan.status
    return compute


def arrival_time_histogram(t_end, status, bins, t_min, t_max, group=None, reduce=False):
    """Histogram (int64 counts) of the arrival times of the surviving rays -- the reduction behind
    pygenray's time-front scatter ``RayFan.plot_time_front``, REF/ray_objects.py:157-222; bins as
    ``np.histogram(t, bins, range=(t_min, t_max))``.  Device tensors go through the HIP kernel behind
    ``pgr_arrival_histogram_device`` (strided views such as ``end[:, 0]`` are read in place); host
    tensors (the gloo rehearsal of the multi-GPU logic) through NumPy itself.  With ``reduce=True``
    `t_end` is this rank's shard and the bins are summed over ranks (all-reduce); with the
    all-gathered fan every rank can histogram locally."""
    bins = int(bins)
    if t_end.is_cuda:
        from . import _lib
        if t_end.dtype != torch.float64 or status.dtype != torch.int32 or t_end.dim() != 1 or status.dim() != 1:
            raise TypeError("arrival_time_histogram: need 1-D float64 times and int32 status")
        n = t_end.shape[0]
        h = torch.empty(bins, dtype=torch.int64, device=t_end.device)
        _lib.arrival_histogram_device(t_end.device.index or 0, t_end.data_ptr(), t_end.stride(0) if n else 1,
                                      status.data_ptr(), status.stride(0) if n else 1, n, t_min, t_max, bins,
                                      h.data_ptr(), torch.cuda.current_stream(t_end.device).cuda_stream)
    else:
        import numpy as np
        t = t_end[(status == 0) & ~torch.isnan(t_end)].numpy()
        h = torch.from_numpy(np.histogram(t, bins=bins, range=(float(t_min), float(t_max)))[0].astype(np.int64))
    if reduce and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    return h


# ------------------------------------------------------------------------------------------------------
# The sharded path as API (REF = /root/reference/src/pygenray): what pygenray does with a process pool over
# rays (REF/launch_rays.py:133-198) and over eigenray brackets (REF/eigenrays.py:122-157), one process per GPU.
# ------------------------------------------------------------------------------------------------------
def _rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _fan_conventions(launch_angles):
    """ODE launch angles and the angles a RayFan stores, as shoot_rays has them (Q1: REF/launch_rays.py:67,94,
    180,251,318): fewer than 70 angles take the serial branch, whose double sign flip integrates +user."""
    if type(launch_angles) is list:
        launch_angles = np.array(launch_angles)
    neg = -np.asarray(launch_angles, dtype=float)
    if len(neg) < 70:
        ode = -neg
        return ode, ode
    return neg, -neg


def hip_end_state_compute(environment, flatearth, rtol=1e-9, terminate_backwards=True, device=None, atol=1e-6,
                          max_steps=1_000_000):
    """`compute(y0_local, x0, x1) -> packed end records [n, 5] on the device` through the HIP fan
    (PGR_PACKED_END: the kernel writes the 40-byte records the all-gather ships); buffers are cached on the
    environment's device handle and grow only."""
    import torch as _torch
    from .device_fan import cached_end_state_fan
    from .launch_rays import _device_env

    def compute(y0_local, x0, x1, backwards, n_pad):
        dev = _torch.cuda.current_device() if device is None else int(device)
        env, _ = _device_env(environment, flatearth, backwards, dev)
        fan = cached_end_state_fan(env, y0_local, x0, x1, rtol=rtol, atol=atol, terminate_backwards=terminate_backwards,
                                   max_steps=max_steps, packed_end=True, n_pad=n_pad)
        fan.run()
        return fan.records
    return compute


def shoot_rays_sharded(source_depth, source_range, launch_angles, receiver_range, environment, rtol=1e-9,
                       terminate_backwards=True, debug=False, flatearth=True, group=None, device=None,
                       compute=None, return_all=False):
    """``shoot_rays`` over all ranks of `group`, end states only: launch angles are dealt to the ranks in a strided
    fashion, every rank integrates its shard (HIP fan, the kernel writes the packed end records) and ONE all-gather
    puts the whole fan's end states, in launch-angle order, on every rank -- the input of the eigenray bracketing
    (REF/eigenrays.py:65-79) and of the arrival-time histogram.  Returns a ``RayFan`` with ONE column (the state at
    ``receiver_range``; stored convention z -> -z, p -> -p; dropped rays have vanished, Q12), identical on every
    rank and identical to the last column of the single-process ``shoot_rays`` fan.  Trajectories stay where they
    are computed: shoot the rays you want to look at with ``shoot_rays`` / ``shoot_ray``.

    ``compute(y0_local, x0, x1, backwards, n_pad) -> records [n_pad, 5]`` replaces the HIP fan in the gloo
    rehearsals of the CPU test suite."""
    from .launch_rays import _report_drops
    from .environment import _unpack_envi, _check_monotone, _mirror_envi_arrays
    from .host_physics import bilinear_interp
    from .ray_objects import RayFan
    rank, world = _rank_world(group)
    ode, stored = _fan_conventions(launch_angles)
    n = len(ode)
    backwards = receiver_range < source_range
    cin, cpin, rin, zin, depths, depth_ranges, bottom_angles = _unpack_envi(environment, flatearth=flatearth)
    _check_monotone(rin, zin, depth_ranges)
    if backwards:
        cin, cpin, rin, depths, depth_ranges, bottom_angles = _mirror_envi_arrays(cin, cpin, rin, depths, depth_ranges, bottom_angles)
    x0, x1 = (-source_range, -receiver_range) if backwards else (source_range, receiver_range)
    if not (x0 < x1):
        raise IndexError("list index out of range")    # as shoot_rays (REF/launch_rays.py:404)
    c = bilinear_interp(x0, source_depth, rin, zin, cin)   # REF/launch_rays.py:140-144
    idx = shard_indices(n, rank, world)
    y0 = np.zeros((len(idx), 3))
    y0[:, 1] = source_depth
    y0[:, 2] = np.sin(np.radians(ode[idx])) / c
    n_pad = (n + world - 1) // world
    if compute is None:
        compute = hip_end_state_compute(environment, flatearth, rtol=rtol, terminate_backwards=terminate_backwards,
                                        device=device)
    records = compute(y0, x0, x1, backwards, n_pad)
    end, nb, ns, st = start_all_gather_records(records, n, group=group).finish()
    end, nb, ns, st = (t.cpu().numpy() for t in (end, nb, ns, st))
    if rank == 0:
        _report_drops(st, debug)
    keep = st == 0
    M = int(keep.sum())
    fan = RayFan.from_arrays(stored[keep], np.full((M, 1), float(receiver_range)), end[keep, 0:1].copy(),
                             -end[keep, 1:2], -end[keep, 2:3], nb[keep].astype(np.int64), ns[keep].astype(np.int64),
                             np.full(M, source_depth))
    if return_all:
        return fan, dict(end=end, n_bott=nb, n_surf=ns, status=st)
    return fan


def _collective_device(group=None):
    """Where the tensors of a collective on `group` live: RCCL ("nccl") moves device memory, gloo host memory."""
    if dist.is_available() and dist.is_initialized() and "nccl" in str(dist.get_backend(group)):
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _all_gather_rows(local, group=None):
    """All-gather one float64 array of the SAME shape on every rank -> ndarray [world, *shape] (one
    ``all_gather_into_tensor``; without an initialised process group: the array itself, world = 1)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    rank, world = _rank_world(group)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local[None]
    dev = _collective_device(group)
    send = torch.from_numpy(local).to(dev)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=torch.float64, device=dev)
    if send.numel():      # (the shape is the same on every rank, so every rank takes the same branch)
        dist.all_gather_into_tensor(out, send, group=group)
    return out.view((world,) + tuple(local.shape)).cpu().numpy()


def find_eigenrays_sharded(rays, receiver_depths, source_depth, source_range, receiver_range, num_range_save,
                           environment, ztol=1, max_iter=20, group=None, refine=None, **kwargs):
    """``find_eigenrays`` over all ranks of `group` (REF/eigenrays.py:11-203).  `rays` is the gathered fan
    (``shoot_rays_sharded`` or any ``RayFan``), identical on every rank: every rank brackets on the WHOLE fan exactly
    as the reference does (brackets across shard edges and across dropped rays included), the brackets of a receiver
    depth are dealt to the ranks (bracket k -> rank k mod world; the reference maps them to a process pool,
    REF/eigenrays.py:122-157), each rank runs the device-resident false-position loop on its own
    (``pgr_eigen_refine``) and shoots its eigenrays with trajectories, and one all-gather of the (small) results
    gives every rank the same ``EigenRays`` -- equal to the single-process result.

    ``refine(z1s, z2s, th1s, th2s, receiver_depth) -> (found, th, r, T, Z, P, nb, ns)`` replaces the HIP refinement
    in the CPU rehearsals."""
    from .eigenrays import _find_eigenrays, _regula_falsi_batch
    rank, world = _rank_world(group)
    S = int(num_range_save)
    together = refine is None     # (the HIP refinement takes a receiver depth per bracket; a stand-in gets one depth per call)
    if refine is None:
        def refine(z1s, z2s, th1s, th2s, receiver_depth):
            return _regula_falsi_batch(z1s, z2s, th1s, th2s, receiver_depth, source_depth, source_range, receiver_range,
                                       num_range_save, environment, ztol, max_iter, kwargs)

    def refine_dealt(z1s, z2s, th1s, th2s, receiver_depth):
        # Bracket k belongs to rank k mod world -- every rank knows every rank's share, so only RESULTS travel, as two
        # tensor collectives of fixed-shape float64 records (no pickled objects: under "nccl" an object gather is bytes
        # through device tensors plus a size exchange per rank):
        #   1. [n_pad, 4] per rank = (found, launch angle, n_bott, n_surf) of its brackets, n_pad = ceil(nbk / world);
        #   2. [f_max, 3, S] per rank = (T, z, p) of the eigenrays it found, f_max = the largest per-rank count (known to
        #      everybody from 1.), plus one row for the save grid.
        nbk = len(z1s)
        n_pad = (nbk + world - 1) // world
        mine = np.arange(rank, nbk, world)
        rec = np.zeros((n_pad, 4))
        Tm = Zm = Pm = np.zeros((0, S))
        r = np.linspace(source_range, receiver_range, S)
        if len(mine):
            rd_m = np.asarray(receiver_depth)[mine] if np.ndim(receiver_depth) else receiver_depth
            found_m, th_m, r_m, T_m, Z_m, P_m, nb_m, ns_m = refine(z1s[mine], z2s[mine], th1s[mine], th2s[mine], rd_m)
            found_m = np.asarray(found_m, bool)
            rec[:len(mine), 0] = found_m
            rec[:len(mine), 1] = np.asarray(th_m, float)
            rec[:len(mine), 2] = np.asarray(nb_m, float)      # (bounce counts: exact in a double)
            rec[:len(mine), 3] = np.asarray(ns_m, float)
            Tm, Zm, Pm = np.asarray(T_m)[found_m], np.asarray(Z_m)[found_m], np.asarray(P_m)[found_m]
            if r_m is not None:
                r = np.asarray(r_m, float)
        recs = _all_gather_rows(rec, group)                              # [world, n_pad, 4]
        n_found = (recs[:, :, 0] != 0).sum(axis=1)
        f_max = int(n_found.max()) if world else 0
        pay = np.zeros((f_max * 3 + 1, S))
        nf = len(Tm)
        pay[0:nf], pay[f_max:f_max + nf], pay[2 * f_max:2 * f_max + nf] = Tm, Zm, Pm
        pay[3 * f_max] = r
        pays = _all_gather_rows(pay, group)                              # [world, 3 f_max + 1, S]
        found = np.zeros(nbk, bool); th = np.zeros(nbk)
        T = np.zeros((nbk, S)); Z = np.zeros((nbk, S)); P = np.zeros((nbk, S))
        nb = np.zeros(nbk, np.int64); ns = np.zeros(nbk, np.int64)
        for q in range(world):
            m = np.arange(q, nbk, world)
            f = recs[q, :len(m), 0] != 0
            found[m] = f
            th[m] = recs[q, :len(m), 1]
            hit = m[f]
            nb[hit] = np.rint(recs[q, :len(m), 2][f]).astype(np.int64)
            ns[hit] = np.rint(recs[q, :len(m), 3][f]).astype(np.int64)
            k = len(hit)
            T[hit], Z[hit], P[hit] = pays[q, 0:k], pays[q, f_max:f_max + k], pays[q, 2 * f_max:2 * f_max + k]
            if k:
                r = pays[q, 3 * f_max]
        return found, th, r, T, Z, P, nb, ns

    return _find_eigenrays(rays, receiver_depths, source_depth, S, environment, refine_dealt, together=together)


def arrival_histogram_sharded(source_depth, source_range, launch_angles, receiver_range, environment, bins, t_min,
                              t_max, rtol=1e-9, terminate_backwards=True, flatearth=True, group=None, device=None,
                              compute=None):
    """Arrival-time histogram of a fan sharded over the ranks (BASELINE configs[4]): every rank integrates its strided
    shard and bins ITS rays on the device (``pgr_arrival_histogram_device``, straight from the packed end records),
    ONE all-reduce (sum) of the int64 counts puts the whole fan's histogram on every rank -- no ray data travels.
    Returns ``(counts [bins] int64 ndarray, edges [bins + 1])`` with ``np.histogram``'s bin rule."""
    from .environment import _unpack_envi, _check_monotone, _mirror_envi_arrays
    from .host_physics import bilinear_interp
    rank, world = _rank_world(group)
    ode, _ = _fan_conventions(launch_angles)
    n = len(ode)
    backwards = receiver_range < source_range
    cin, cpin, rin, zin, depths, depth_ranges, bottom_angles = _unpack_envi(environment, flatearth=flatearth)
    _check_monotone(rin, zin, depth_ranges)
    if backwards:
        cin, cpin, rin, depths, depth_ranges, bottom_angles = _mirror_envi_arrays(cin, cpin, rin, depths, depth_ranges, bottom_angles)
    x0, x1 = (-source_range, -receiver_range) if backwards else (source_range, receiver_range)
    c = bilinear_interp(x0, source_depth, rin, zin, cin)
    idx = shard_indices(n, rank, world)
    y0 = np.zeros((len(idx), 3))
    y0[:, 1] = source_depth
    y0[:, 2] = np.sin(np.radians(ode[idx])) / c
    if compute is None:
        compute = hip_end_state_compute(environment, flatearth, rtol=rtol, terminate_backwards=terminate_backwards,
                                        device=device)
    records = compute(y0, x0, x1, backwards, len(idx))
    ints = records[:, 3:5].view(torch.int32)
    h = arrival_time_histogram(records[:len(idx), 0], ints[:len(idx), 2], bins, t_min, t_max, group=group, reduce=True)
    return h.cpu().numpy(), np.linspace(float(t_min), float(t_max), int(bins) + 1)


__all__ = ["shard_indices", "shoot_rays_sharded", "find_eigenrays_sharded", "arrival_histogram_sharded",
           "shoot_fan_sharded", "all_gather_fan", "start_all_gather_fan", "start_all_gather_records",
           "arrival_time_histogram", "hip_compute", "hip_end_state_compute"]
