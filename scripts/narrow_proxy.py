"""Round 6, verdict item 1, the proxy BEFORE any kernel work: is there ROOM in the 1e5-ray fan for more, narrower waves?

A "wide" instance (one ray on a quad of lanes, T / z / p in parallel) turns a steep 64-ray packet into four waves of 16
rays.  Whatever it saves per trip, the fan only gains if the chip can take the extra waves without slowing the steep ones
down.  That part needs NO new kernel: the existing one skips rays whose y0 is NaN (PGR_SKIP_NAN_Y0, the eigenray search's
parked brackets), so a y0 array in which the K steepest packets are re-dealt as `64 / width` packets of `width` real rays +
NaN padding IS the narrow-wave fan -- same rays, same bits, more waves.  Timed here for K = 0 ... and widths 32 / 16 / 8:
the fan's kernel time with and without trajectories, next to the lone steepest packet at each width.

    python scripts/narrow_proxy.py [--K 0 8 16 32 64 96 128] [--width 16 32] [--reps 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from pygenray_amd import _lib
from pygenray_amd.device_fan import DeviceFan, fan_y0
from helpers import munk_arrays

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=100000)
ap.add_argument("--K", type=int, nargs="*", default=[0, 8, 16, 32, 64, 96, 128, 192])
ap.add_argument("--width", type=int, nargs="*", default=[16, 32])
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--S", type=int, default=1001)
ap.add_argument("--slope", type=float, default=0.0)
ap.add_argument("--lib", default=None)
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)

arrs = munk_arrays(1000e3, nr=(101 if a.slope else 100), sofar_slope=a.slope)
env = _lib.EnvHandle(*arrs)
theta = np.linspace(-20, 20, a.rays)
y0 = fan_y0(arrs, 1000.0, 0.0, -theta)
N = len(y0)
P = (N + 63) // 64
cost = np.array([np.abs(y0[64 * p:64 * p + 64, 2]).max() for p in range(P)])
order = np.argsort(-cost, kind="stable")


def redeal(K, width):
    """y0 with the K costliest 64-ray packets re-dealt as packets of `width` rays + NaN padding (in front), the rest as it was."""
    if K == 0:
        return y0, np.arange(N)
    top = set(order[:K].tolist())
    rows, src = [], []
    for p in order[:K]:
        blk = np.arange(64 * p, min(64 * p + 64, N))
        for o in range(0, len(blk), width):
            part = blk[o:o + width]
            pad = np.full((64, 3), np.nan)
            pad[:, 0] = 0.0
            pad[:len(part)] = y0[part]
            rows.append(pad)
            s = np.full(64, -1)
            s[:len(part)] = part
            src.append(s)
    rest = np.concatenate([np.arange(64 * p, min(64 * p + 64, N)) for p in range(P) if p not in top])
    rows.append(y0[rest])
    src.append(rest)
    return np.concatenate(rows), np.concatenate(src)


def time_fan(yy, save, reps=a.reps):
    fan = DeviceFan(env, yy, 0.0, 1000e3, a.S, save=save, sample_major=save)
    fan.flags |= _lib.PGR_SKIP_NAN_Y0
    fan.run(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fan.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    st = fan.status.cpu().numpy()
    steps = int(fan.n_steps[torch.from_numpy(st != 8).to(fan.dev)].sum(dtype=torch.int64).item())
    end = fan.end.cpu().numpy()
    del fan
    torch.cuda.empty_cache()
    return min(ts), float(np.median(ts)), steps, end, st


print(f"# {a.rays} rays, {P} packets; build: {_lib.build_info()}", flush=True)
base = {}
for save in (False, True):
    ms, med, steps, end, st = time_fan(y0, save)
    base[save] = (ms, steps, end)
    print(f"baseline save={int(save)}: kernel {ms:.3f} ms (median {med:.3f}), {steps} steps, waves {P}", flush=True)

# the lone steepest packet at each width (one wave on the chip)
top = np.arange(64 * order[0], min(64 * order[0] + 64, N))
for width in (64, 32, 16, 8, 4):
    pad = np.full((64, 3), np.nan); pad[:, 0] = 0.0
    pad[:width] = y0[top[:width]]
    for save in (False, True):
        ms, med, steps, _, _ = time_fan(pad, save, reps=3)
        print(f"lone steepest packet, {width:2d} rays in the wave, save={int(save)}: {ms:.3f} ms", flush=True)

for width in a.width:
    for K in a.K:
        if K == 0:
            continue
        yy, src = redeal(K, width)
        waves = (len(yy) + 63) // 64
        if waves > 2048:
            print(f"width {width} K {K}: {waves} waves > 2048 slots, skipped")
            continue
        for save in (False, True):
            ms, med, steps, end, st = time_fan(yy, save)
            real = src >= 0
            same = np.array_equal(end[real], base[save][2][src[real]], equal_nan=True)
            print(f"width {width:2d} K {K:4d} save={int(save)}: kernel {ms:.3f} ms (median {med:.3f}) vs {base[save][0]:.3f}  "
                  f"waves {waves}  steps equal {steps == base[save][1]}  end states bit-equal {same}", flush=True)
