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
    from .device_fan import DeviceFan

    def compute(y0_local):
        fan = DeviceFan(env_handle, y0_local, source_range, receiver_range, 1, rtol=rtol, atol=atol,
                        terminate_backwards=terminate_backwards, save=False, max_steps=max_steps)
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
