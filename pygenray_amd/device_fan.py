"""Device-resident fan launches: inputs and outputs stay in HBM.

PyTorch is plumbing only here (device memory + the current HIP stream): the tensors'
``data_ptr()`` go straight into ``pgr_shoot_fan_device`` (include/pgr.h).  Used by bench.py
(whole-job throughput with inputs already resident) and by the multi-GPU driver.
"""
import numpy as np
import torch

from . import _lib
from .environment import _unpack_envi
from .host_physics import bilinear_interp


class DeviceFan:
    """Buffers for one fan of N rays on ``device`` and a ``run()`` that enqueues the kernel."""

    def __init__(self, env_handle, y0, source_range, receiver_range, num_range_save, rtol=1e-9,
                 atol=1e-6, terminate_backwards=True, save=True, sample_major=False,
                 max_steps=1_000_000, exact_bisection=False, exact_samples=False, packed_end=False,
                 n_pad=None, sample_blocked=None):
        """``sample_blocked``: None (default) = the layout that suits the environment -- sample-major trajectory fans of
        HBM-table environments (range-dependent tables, depth grids too large for the LDS) are written sample-blocked,
        ``[ceil(S/4)][N][4]`` (row stores leave L2 half-written there: 2.3 x the sample bytes reach HBM; blocked 1.2 x);
        everything else in rows ``[S][N]``.  ``rows(t)`` is the (S, N) form either way.  True / False force the choice."""
        self.env = env_handle
        dev = torch.device("cuda", env_handle.device)
        self.dev = dev
        y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(-1, 3)
        self.N, self.S = len(y0), int(num_range_save)
        self.x0, self.x1 = float(source_range), float(receiver_range)
        self.rtol, self.atol, self.max_steps = float(rtol), float(atol), int(max_steps)
        if sample_blocked is None:
            sample_blocked = bool(save and sample_major and not exact_samples and getattr(env_handle, "blocked_layout", False))
        self.flags = (_lib.PGR_TERMINATE_BACKWARDS if terminate_backwards else 0) | \
            (_lib.PGR_SAMPLE_MAJOR if sample_major else 0) | _lib.PGR_SAVE_LINSPACE | \
            (_lib.PGR_EXACT_BISECTION if exact_bisection else 0) | \
            (_lib.PGR_EXACT_SAMPLES if exact_samples else 0) | (_lib.PGR_PACKED_END if packed_end else 0) | \
            (_lib.PGR_SAMPLE_BLOCKED if sample_blocked else 0)
        if sample_blocked and not (save and sample_major):
            raise ValueError("sample_blocked goes with save=True, sample_major=True")
        self.save, self.sample_major, self.packed_end, self.sample_blocked = save, sample_major, packed_end, sample_blocked
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.y0 = torch.from_numpy(y0).to(dev)
        self.r_save = torch.from_numpy(np.linspace(self.x0, self.x1, self.S)).to(dev)
        if save:
            shape = (self.S, self.N) if sample_major else (self.N, self.S)
            if sample_blocked:
                # PGR_SAMPLE_BLOCKED (include/pgr.h): [ceil(S/4)][N][4]; `rows(t)` gives the (S, N) view of such a tensor
                shape = ((self.S + 3) // 4, self.N, 4)
            self.T = torch.empty(shape, **f64)
            self.Z = torch.empty(shape, **f64)
            self.P = torch.empty(shape, **f64)
        else:
            self.T = self.Z = self.P = None
        if packed_end:
            # [n_pad][5] end records (PGR_PACKED_END), ready for the all-gather: rows beyond N stay
            # zero (valid = 0); `end` is the (N, 3) view of the state columns
            self.records = torch.zeros((max(int(n_pad or 0), self.N), 5), **f64)
            self.end = self.records[:self.N, 0:3]
        else:
            self.records = None
            self.end = torch.empty((self.N, 3), **f64)
        self.n_bott = torch.empty(self.N, **i32)
        self.n_surf = torch.empty(self.N, **i32)
        self.status = torch.empty(self.N, **i32)
        self.n_steps = torch.empty(self.N, **i32)
        self.n_rej = torch.empty(self.N, **i32)
        # (the full-capacity buffers behind the views set_y0 hands out)
        self.y0_buf, self.records_buf, self.end_buf = self.y0, self.records, (None if packed_end else self.end)
        self.nb_buf, self.ns_buf, self.st_buf = self.n_bott, self.n_surf, self.status
        self.nsteps_buf, self.nrej_buf = self.n_steps, self.n_rej
        self.n_pad_min = int(n_pad or 0)

    def rows(self, t):
        """(S, N) form of a trajectory tensor (sample-major fans): row j = sample j of every ray.  The tensor itself, or --
        for a sample_blocked fan, whose [ceil(S/4)][N][4] buffer has no 2-D strided view -- an un-blocked COPY
        (`t.permute(0, 2, 1)` is the (S/4, 4, N) view without a copy)."""
        if not self.sample_blocked:
            return t
        return t.permute(0, 2, 1).flatten(0, 1)[:self.S]

    def set_y0(self, y0):
        """New initial states for the same buffers (a fan of the same size or smaller: the per-ray outputs are
        views of the first N entries): what a sharded fan or an eigenray search re-uses across calls instead of
        nine fresh allocations per fan."""
        y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(-1, 3)
        n = len(y0)
        cap = self.y0_buf.shape[0]
        if n > cap or self.save:
            raise ValueError("set_y0: the fan's buffers hold %d rays (end-state fans only)" % cap)
        self.N = n
        self.y0_buf[:n].copy_(torch.from_numpy(y0))
        self.y0 = self.y0_buf[:n]
        if self.packed_end:
            self.records = self.records_buf[:max(self.n_pad_min, n)]
            self.records[n:].zero_()
            self.end = self.records[:n, 0:3]
        else:
            self.end = self.end_buf[:n]
        self.n_bott, self.n_surf, self.status = self.nb_buf[:n], self.ns_buf[:n], self.st_buf[:n]
        self.n_steps, self.n_rej = self.nsteps_buf[:n], self.nrej_buf[:n]

    def run(self):
        """Enqueue one pass of the hot path on torch's current stream (asynchronous)."""
        ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.env.shoot_fan_device(ptr(self.y0), self.N, self.x0, self.x1, ptr(self.r_save), self.S,
                                  self.rtol, self.atol, self.flags, self.max_steps, ptr(self.T),
                                  ptr(self.Z), ptr(self.P),
                                  ptr(self.records if self.packed_end else self.end), ptr(self.n_bott),
                                  ptr(self.n_surf), ptr(self.status), ptr(self.n_steps),
                                  ptr(self.n_rej), stream)

    def ray_steps(self):
        return int(self.n_steps.sum(dtype=torch.int64).item())


def cached_end_state_fan(env_handle, y0, source_range, receiver_range, rtol=1e-9, atol=1e-6,
                         terminate_backwards=True, max_steps=1_000_000, packed_end=False, n_pad=None):
    """An end-state-only DeviceFan for these rays whose buffers live on the environment handle and grow only:
    the many fans of a sharded search re-use one set of allocations."""
    y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(-1, 3)
    key = ("end_state_fan", bool(packed_end))
    cache = env_handle.__dict__.setdefault("_fan_cache", {})
    fan = cache.get(key)
    need = max(len(y0), int(n_pad or 0))
    if fan is None or fan.y0_buf.shape[0] < need or (packed_end and fan.records_buf.shape[0] < need):
        cap = max(need, 64)
        fan = DeviceFan(env_handle, np.zeros((cap, 3)), source_range, receiver_range, 1, rtol=rtol, atol=atol,
                        terminate_backwards=terminate_backwards, save=False, max_steps=max_steps,
                        packed_end=packed_end, n_pad=cap)
        cache[key] = fan
    fan.x0, fan.x1 = float(source_range), float(receiver_range)
    fan.rtol, fan.atol, fan.max_steps = float(rtol), float(atol), int(max_steps)
    fan.flags = (fan.flags & ~_lib.PGR_TERMINATE_BACKWARDS) | (_lib.PGR_TERMINATE_BACKWARDS if terminate_backwards else 0)
    fan.n_pad_min = int(n_pad or 0)
    fan.set_y0(y0)
    return fan


def fan_y0(arrays, source_depth, source_range, ode_angles_deg):
    """y0 = [0, z_s, sin(theta)/c(x_s, z_s)] per ray (REF/launch_rays.py:140-144)."""
    cin, _, rin, zin = arrays[:4]
    c = bilinear_interp(source_range, source_depth, rin, zin, cin)
    ang = np.asarray(ode_angles_deg, dtype=float)
    y0 = np.zeros((len(ang), 3))
    y0[:, 1] = source_depth
    y0[:, 2] = np.sin(np.radians(ang)) / c
    return y0


def env_handle_from(environment, flatearth=False, device=0):
    arrs = _unpack_envi(environment, flatearth=flatearth)
    return _lib.EnvHandle(*arrs, device=device), arrs
