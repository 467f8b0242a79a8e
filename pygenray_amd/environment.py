"""Ocean environment front end: the table *producer* side of the hot path.

Mirrors ``pygenray.environment`` (REF = /root/reference/src/pygenray):
``OceanEnvironment2D`` ctor defaults/validation (REF/environment.py:49-119), the WGS-84
flat-earth transform ``eflat`` / ``eflatinv`` (REF/environment.py:371-453) and ``munk_ssp``
(REF/environment.py:218-236), plus ``_unpack_envi`` / ``_mirror_envi_arrays``
(REF/launch_rays.py:684-742) producing the 7-array contract that is uploaded to HBM once
per environment (instead of once per ray, REF/launch_rays.py:253-255).
"""
import numpy as np
import scipy.interpolate

from .xr_lite import DataArray, is_dataarray, coord_values


def munk_ssp(z, sofar_depth=1300, eps=0.00737):
    """Munk sound-speed profile (REF/environment.py:218-236)."""
    zh = 2 * (z - sofar_depth) / sofar_depth
    return 1500 * (1 + eps * (zh - 1 + np.exp(-zh)))


def _earth_radius(lat):
    # WGS-84 radius used by the flat-earth maps (REF/environment.py:382-395)
    wgsa = 6378137.0
    wgsb = 6356752.314
    wgsfact = (wgsb / wgsa) ** 4
    wgsa = wgsa * wgsa
    wgsb = wgsb * wgsb
    ll = np.pi * lat / 180.0
    ree1 = wgsa / np.sqrt(wgsa * np.cos(ll) * np.cos(ll) + wgsb * np.sin(ll) * np.sin(ll))
    return ree1 * np.sqrt(np.cos(ll) * np.cos(ll) + wgsfact * np.sin(ll) * np.sin(ll))


def eflat(dep, lat, cs=None):
    """Flat-earth transformation of depths and sound speeds (REF/environment.py:371-401)."""
    if cs is None:
        cs = np.zeros_like(dep)
    re = _earth_radius(lat)
    E = dep / re
    depf = dep * (1.0 + E * (0.50 + E / 3.0))
    csf = cs * (1.0 + E * (1.0 + E))
    return depf, csf


def eflatinv(depf, lat, csf=None):
    """Inverse flat-earth transformation (REF/environment.py:404-453): solves
    eflat(dep) = depf for dep (to 1 mm, as the reference's Ridder solver) by bracketed
    root finding, then un-scales the sound speed."""
    depf = np.reshape(np.asarray(depf, dtype=float), (-1,))
    lat = np.reshape(np.asarray(lat, dtype=float), (-1,))
    csf = np.zeros(depf.shape) if csf is None else np.reshape(np.asarray(csf, dtype=float), (-1,))
    re = _earth_radius(lat)
    lo, hi = depf * 0.1, depf.copy()
    for _ in range(60):  # bisection on a monotone map; 60 halvings >> 1 mm
        mid = 0.5 * (lo + hi)
        too_deep = eflat(mid, lat)[0] > depf
        hi = np.where(too_deep, mid, hi)
        lo = np.where(too_deep, lo, mid)
    dep = 0.5 * (lo + hi)
    E = dep / re
    cs = csf / (1.0 + E * (1.0 + E))
    return dep, cs


def flat_earth_c(c, verbose=False, n_cpus=None, chunk_size=None):
    """Flat-earth transformed sound-speed slice, latitude varying along the range (REF/environment.py:239-303): for
    every range column ``eflat(depth, lat_i, c[:, i])`` gives transformed depths and speeds, and the speeds are
    interpolated back -- linearly, NaN outside the transformed span, as ``xarray.DataArray.interp`` does -- onto the
    ORIGINAL depth grid.  `c`: dims ('depth', 'range') or ('range', 'depth') with coordinates depth, range and
    ``lat`` (one latitude per range).  ``n_cpus`` / ``chunk_size`` (the reference's process pool) are accepted and
    ignored: the columns are a few NumPy calls each.  Returns a DataArray laid out like the reference's result
    (dims ('range', 'depth'), coordinates range, depth, lat)."""
    _as_dataarray(c, "c")
    if c.ndim != 2 or "depth" not in c.dims or "range" not in c.dims:
        raise ValueError("c must be 2D with dimensions 'depth' and 'range'.")
    coords = c.coords
    if "lat" not in coords:
        raise ValueError("c must have a 'lat' coordinate along 'range'.")
    v = np.asarray(c.values, dtype=float)
    if tuple(c.dims) == ("depth", "range"):
        v = v.T
    z = coord_values(c, "depth")
    r = coord_values(c, "range")
    lat = np.asarray(getattr(coords["lat"], "values", coords["lat"]), dtype=float).reshape(-1)
    if len(lat) != len(r):
        raise ValueError("the 'lat' coordinate must have one entry per range.")
    out = np.empty_like(v)
    for i in range(len(r)):
        depf, cf = eflat(z, lat[i], v[i])
        col = np.interp(z, depf, cf)
        col[(z < depf[0]) | (z > depf[-1])] = np.nan      # xarray's interp: NaN outside the source coordinates
        out[i] = col
    if verbose:
        print(f"Processed {len(r)} range points")
    return DataArray(out, dims=["range", "depth"], coords={"range": r, "depth": z, "lat": lat})


def _as_dataarray(obj, what):
    if not is_dataarray(obj):
        raise TypeError(f"{what} must be an xarray DataArray.")
    return obj


class OceanEnvironment2D:
    """Ocean environment specification (2D); same constructor, defaults, validation errors
    and attributes as ``pygenray.OceanEnvironment2D`` (REF/environment.py:14-119).

    ``sound_speed`` / ``bathymetry`` may be real ``xarray.DataArray`` objects or
    ``pygenray_amd.DataArray`` (no xarray needed).
    """

    def __init__(self, sound_speed=None, bathymetry=None, lat=35, flat_earth_transform=True,
                 verbose=False):
        self.latitude = lat
        if sound_speed is None:
            z = np.arange(0, 6000, 1)
            c_munk = munk_ssp(z)
            sound_speed = DataArray(np.array([c_munk] * 100), dims=["range", "depth"],
                                    coords={"depth": z, "range": np.linspace(0, 100e3, 100)})
        else:
            _as_dataarray(sound_speed, "sound_speed")
            if sound_speed.ndim not in [1, 2]:
                raise ValueError("sound_speed must be 1D or 2D.")
            if "depth" not in sound_speed.dims:
                raise ValueError("sound_speed must have a 'depth' dimension.")
            if sound_speed.ndim == 2 and "range" not in sound_speed.dims:
                raise ValueError("2D sound_speed must have a 'range' dimension.")
        if bathymetry is None:
            # NB the reference's default is a 4500 -> 4900 m slope (Q11)
            bathymetry = DataArray(np.linspace(4500, 4900, 100), dims=["range"],
                                   coords={"range": np.linspace(0, 100e3, 100)})
        else:
            _as_dataarray(bathymetry, "bathymetry")
            if bathymetry.ndim != 1:
                raise ValueError("bathymetry must be 1D.")
            if "range" not in bathymetry.dims:
                raise ValueError("bathymetry must have a 'range' dimension.")

        self.sound_speed = sound_speed
        self.dcdz = np.asarray(sound_speed.differentiate("depth").values)
        self.bathymetry = bathymetry
        self._cache = {}

        if flat_earth_transform:
            self.flat_earth_transform(lat=lat)

        # bottom slope from the UNtransformed bathymetry (Q10), degrees
        b_r = coord_values(self.bathymetry, "range")
        bottom_slope = np.gradient(np.asarray(self.bathymetry.values, dtype=float), b_r)
        self.bottom_angle = np.degrees(np.arctan(bottom_slope))
        self.bottom_angle_interp = scipy.interpolate.interp1d(b_r, self.bottom_angle, kind="cubic")

    # ---- tables in (range, depth) order ----
    @staticmethod
    def _range_depth(da):
        v = np.asarray(da.values, dtype=float)
        if da.ndim != 2:
            raise ValueError("a 2D (range, depth) sound speed is required to shoot rays")  # Q14
        if tuple(da.dims) == ("depth", "range"):
            v = v.T
        return np.ascontiguousarray(v), coord_values(da, "range"), coord_values(da, "depth")

    def flat_earth_transform(self, lat):
        """Single-latitude earth flattening (REF/environment.py:121-154)."""
        c, r, z = self._range_depth(self.sound_speed)
        depf, _ = eflat(z, lat, c[0])
        cf = np.array([eflat(z, lat, row)[1] for row in c])
        self.sound_speed_fe = DataArray(cf, dims=["range", "depth"], coords={"range": r, "depth": depf})
        bathy_flat, _ = eflat(np.asarray(self.bathymetry.values, dtype=float), lat)
        self.bathymetry_fe = DataArray(bathy_flat, dims=["range"],
                                       coords={"range": coord_values(self.bathymetry, "range")})
        self._cache = {}

    def flat_earth_transform_rd(self):
        """Earth flattening computed for each range / latitude independently (REF/environment.py:156-173):
        ``sound_speed`` must carry a ``lat`` coordinate along ``range``; the bathymetry is left as it is (as in
        the reference)."""
        c_fe = flat_earth_c(self.sound_speed, verbose=False)
        self.sound_speed_fe = c_fe
        self.dcdz = c_fe.differentiate("depth")
        self.bathymetry_fe = self.bathymetry.copy(deep=True)
        self._cache = {}

    def plot(self, **kwargs):
        """2D slice of the environment (REF/environment.py:171-215)."""
        from matplotlib import pyplot as plt
        c, r, z = self._range_depth(self.sound_speed)
        add_colorbar = kwargs.pop("add_colorbar", True)
        kw = dict(cmap="viridis")
        kw.update(kwargs)
        m = plt.pcolormesh(r, z, c.T, **kw)
        if add_colorbar:
            plt.colorbar(m, label="sound speed [m/s]")
        b_r = coord_values(self.bathymetry, "range")
        plt.fill_between(b_r, np.asarray(self.bathymetry.values), 50000, color="#aaaaaa", alpha=1, lw=0)
        plt.xlabel("range [m]")
        plt.ylabel("depth [m]")
        plt.ylim(z.max(), z.min())


def _unpack_envi(environment, flatearth=True):
    """The 7-array contract (REF/launch_rays.py:717-742), cached on the environment:
    cin, cpin (= d c / d depth by np.gradient, as xarray's differentiate), rin, zin, depths,
    depth_ranges, bottom_angles."""
    cache = getattr(environment, "_cache", None)
    src = (getattr(environment, "sound_speed_fe", None) if flatearth else environment.sound_speed,
           getattr(environment, "bathymetry_fe", None) if flatearth else environment.bathymetry)
    key = ("arrays", bool(flatearth), id(src[0]), id(src[1]))
    if cache is not None and key in cache:
        return cache[key]
    if flatearth:
        if not hasattr(environment, "sound_speed_fe"):
            raise Exception(
                "Flat earth transformation has not been applied. Set `flat_earth_transform=True` "
                "when creating the OceanEnvironment2D object.")
        ss, ba = environment.sound_speed_fe, environment.bathymetry_fe
    else:
        ss, ba = environment.sound_speed, environment.bathymetry
    cin, rin, zin = OceanEnvironment2D._range_depth(ss)
    cpin = np.gradient(cin, zin, axis=1, edge_order=1)
    depths = np.asarray(ba.values, dtype=float)
    depth_ranges = coord_values(ba, "range")
    bottom_angles = np.asarray(environment.bottom_angle, dtype=float)
    out = (cin, cpin, rin, zin, depths, depth_ranges, bottom_angles)
    if cache is not None:
        cache[key] = out
    return out


def _mirror_envi_arrays(cin, cpin, rin, depths, depth_ranges, bottom_angles):
    """x' = -x reflection for backwards shots (REF/launch_rays.py:684-714)."""
    return (np.ascontiguousarray(cin[::-1, :]), np.ascontiguousarray(cpin[::-1, :]),
            -rin[::-1], np.ascontiguousarray(depths[::-1]), -depth_ranges[::-1],
            -bottom_angles[::-1])


def _check_monotone(rin, zin, depth_ranges):
    # REF/launch_rays.py:79-90 / 258-269
    if not (np.all(np.diff(rin) >= 0)):
        raise Exception("Sound speed range coordinates must be monotonically increasing.")
    if not (np.all(np.diff(zin) >= 0)):
        raise Exception("Sound speed depth coordinates must be monotonically increasing.")
    if not (np.all(np.diff(depth_ranges) >= 0)):
        raise Exception("Bathymetry range coordinates must be monotonically increasing.")


__all__ = ["OceanEnvironment2D", "munk_ssp", "eflat", "eflatinv", "flat_earth_c", "DataArray"]
