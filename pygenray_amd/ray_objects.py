"""Result containers with pygenray's layout and sign convention.

Mirrors ``pygenray.ray_objects`` (REF = /root/reference/src/pygenray): ``Ray``
(REF/ray_objects.py:7-59), ``RayFan`` (REF/ray_objects.py:75-155, 262-430) and ``EigenRays``
(REF/ray_objects.py:433-548).  Stored convention: ``z = -z_ode``, ``p = -p_ode`` (Q3).
``RayFan.from_arrays`` builds a fan straight from the SoA buffers the HIP path returns,
without materialising per-ray Python objects.
"""
import numpy as np


class Ray:
    """Single ray (REF/ray_objects.py:7-59).  ``y`` is ODE-convention [T; z; p] of shape (3, S)."""

    def __init__(self, r, y, n_bottom, n_surface, launch_angle=None, source_depth=None):
        self.r = r
        self.t = y[0, :]
        self.z = -y[1, :]  # negative-z storage convention
        self.p = -y[2, :]
        self.n_bottom = n_bottom
        self.n_surface = n_surface
        if launch_angle is not None:
            self.launch_angle = launch_angle
        if source_depth is not None:
            self.source_depth = source_depth

    def plot(self, **kwargs):
        from matplotlib import pyplot as plt
        plt.plot(self.r, self.z, **kwargs)
        plt.xlabel("time [s]")
        plt.ylabel("depth [m]")
        plt.ylim([self.z.min(), self.z.max()])


class RayFan:
    """Ray fan (REF/ray_objects.py:75-136): ``thetas (M,)``, ``rs/ts/zs/ps (M, S)``,
    ``n_botts/n_surfs/source_depths (M,)``, ``ray_ids (M,)``."""

    def __init__(self, Rays):
        self.thetas = np.array([r.launch_angle for r in Rays])
        self.rs = np.array([r.r for r in Rays])
        self.ts = np.array([r.t for r in Rays])
        self.zs = np.array([r.z for r in Rays])
        self.ps = np.array([r.p for r in Rays])
        self.n_botts = np.array([r.n_bottom for r in Rays])
        self.n_surfs = np.array([r.n_surface for r in Rays])
        self.source_depths = np.array([r.source_depth for r in Rays])
        self.compute_rayids()

    @classmethod
    def from_arrays(cls, thetas, rs, ts, zs, ps, n_botts, n_surfs, source_depths):
        """Stored-convention arrays in, no copies of the (M, S) blocks."""
        self = cls.__new__(cls)
        self.thetas = np.asarray(thetas)
        self.rs, self.ts, self.zs, self.ps = rs, ts, zs, ps
        self.n_botts = np.asarray(n_botts)
        self.n_surfs = np.asarray(n_surfs)
        self.source_depths = np.asarray(source_depths)
        self._ray_ids = None  # built on first access (a million-ray fan pays 0.2 s for the strings)
        return self

    @classmethod
    def from_device(cls, handle, thetas, r, end, n_botts, n_surfs, source_depths):
        """A fan whose trajectories are still in HBM (``_lib.FanHandle``, launched with the stored sign convention):
        ``ts`` / ``zs`` / ``ps`` cross PCIe when they are first read -- each on its own, dropped rays already squeezed
        out on the device -- and are ordinary (M, S) arrays from then on; ``rs`` is a broadcast view of the save grid.
        The per-ray arrays and the end states (``ts_end``, ``zs_end``, ``ps_end``: the last column, what
        ``find_eigenrays`` brackets on, REF/eigenrays.py:65-79) are on the host from the start.  `end` is the
        ODE-convention end state of the surviving rays."""
        self = cls.__new__(cls)
        self.thetas = np.asarray(thetas)
        self._dev = handle
        self._r = np.asarray(r)
        self._end = np.asarray(end)
        self.n_botts = np.asarray(n_botts)
        self.n_surfs = np.asarray(n_surfs)
        self.source_depths = np.asarray(source_depths)
        self._ray_ids = None
        return self

    # ts / zs / ps / rs: plain attributes for a host fan, fetched from the device on first access for a device fan
    def _lazy(name, key):   # noqa: N805  (a property factory, not a method)
        def get(self):
            d = self.__dict__
            if name not in d:
                dev = d.get("_dev")
                if dev is None:
                    raise AttributeError(name[1:])
                d[name] = dev.fetch_samples((key,), compact=True)[key].T     # (M, S) view of the [S][M] block
                if all(k in d for k in ("_ts", "_zs", "_ps")):
                    dev.close()             # everything is on the host: give the HBM back
                    d["_dev"] = None
            return d[name]

        def set_(self, value):
            self.__dict__[name] = value
        return property(get, set_)

    ts = _lazy("_ts", "T")
    zs = _lazy("_zs", "z")
    ps = _lazy("_ps", "p")
    del _lazy

    @property
    def rs(self):
        d = self.__dict__
        if "_rs" not in d:
            if d.get("_r") is None:
                raise AttributeError("rs")
            d["_rs"] = np.broadcast_to(d["_r"], (len(self.thetas), len(d["_r"])))
        return d["_rs"]

    @rs.setter
    def rs(self, value):
        self.__dict__["_rs"] = value

    @property
    def device_resident(self):
        """True while some trajectory array has not been fetched from the GPU yet."""
        return self.__dict__.get("_dev") is not None

    # A device-resident fan pins 3 * N * S * 8 B of HBM (2.4 GB for 1e5 rays x 1001 samples) until ts, zs AND ps have been
    # read, or to_host() / release() is called, or the fan is garbage collected; `rs` of such a fan is a read-only broadcast
    # view of the save grid (np.array(fan.rs) for a private copy).
    def to_host(self):
        """Fetch whatever is still on the GPU and give the HBM back; the fan is a plain host object afterwards."""
        if self.device_resident:
            self.ts, self.zs, self.ps       # noqa: B018  (the third read closes the handle)
        return self

    def release(self):
        """Give the HBM back WITHOUT fetching: trajectory arrays not read so far are gone (reading them raises
        AttributeError); end states, bounce counts and launch angles stay."""
        dev = self.__dict__.get("_dev")
        if dev is not None:
            dev.close()
            self.__dict__["_dev"] = None

    def __getstate__(self):
        """pickle / copy.deepcopy / multiprocessing: the state of a plain host fan (the reference's RayFan is a plain
        object).  A device-resident fan is fetched first -- its handle wraps a device pointer that means nothing in
        another process."""
        self.to_host()
        d = dict(self.__dict__)
        d.pop("_dev", None)
        if d.get("_r") is not None:
            d.pop("_rs", None)              # the broadcast view is rebuilt from the save grid on first access
        return d

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault("_ray_ids", None)

    # the state at receiver_range, stored convention, without touching the trajectories
    @property
    def ts_end(self):
        e = self.__dict__.get("_end")
        return e[:, 0] if e is not None else self.ts[:, -1]

    @property
    def zs_end(self):
        e = self.__dict__.get("_end")
        if e is None:
            return self.zs[:, -1]
        if self.__dict__.get("_zs_end") is None:      # (what find_eigenrays brackets on, once per receiver depth)
            self.__dict__["_zs_end"] = -e[:, 1]
        return self.__dict__["_zs_end"]

    @property
    def ps_end(self):
        e = self.__dict__.get("_end")
        return -e[:, 2] if e is not None else self.ps[:, -1]

    @property
    def ray_ids(self):
        if getattr(self, "_ray_ids", None) is None:
            self.compute_rayids()
        return self._ray_ids

    @ray_ids.setter
    def ray_ids(self, value):
        self._ray_ids = value

    def compute_rayids(self):
        """Ray IDs: number of sign changes of p times sign(theta), 'b' suffix when the ray
        touched a boundary (REF/ray_objects.py:138-155)."""
        if len(self.thetas) == 0:
            self.ray_ids = np.array([], dtype=str)
            return
        ray_ids = np.sum(np.diff(np.sign(self.ps)) != 0, axis=1) * (np.sign(self.thetas))
        b_mask = (self.n_botts == 0) & (self.n_surfs == 0)
        txt = ray_ids.astype(str)  # same text as str(np.float64): '-3.0', '0.0', ...
        self.ray_ids = np.where(b_mask, txt, np.char.add(txt, "b"))

    def __len__(self):
        return len(self.thetas)

    def _ray(self, i):
        # REF/ray_objects.py:385-393: re-negate so Ray() flips back to the stored convention
        return Ray(r=self.rs[i], y=np.array([self.ts[i], -self.zs[i], -self.ps[i]]),
                   n_bottom=self.n_botts[i], n_surface=self.n_surfs[i],
                   launch_angle=self.thetas[i], source_depth=self.source_depths[i])

    def __getitem__(self, key):
        """int -> Ray; slice / index array / boolean mask -> RayFan (REF/ray_objects.py:358-430)."""
        if isinstance(key, (int, np.integer)):
            key = int(key)
            if key < 0:
                key = len(self.thetas) + key
            if key < 0 or key >= len(self.thetas):
                raise IndexError(
                    f"Index {key} is out of bounds for RayFan with {len(self.thetas)} rays")
            return self._ray(key)
        if isinstance(key, slice):
            idx = np.arange(len(self.thetas))[key]
        else:
            idx = np.asarray(key)
            if idx.dtype == bool:
                idx = np.where(idx)[0]
        if idx.ndim == 0:
            idx = idx.reshape(1)
        elif idx.ndim != 1:
            raise ValueError("Invalid indexing array shape")
        return RayFan.from_arrays(self.thetas[idx], self.rs[idx], self.ts[idx], self.zs[idx],
                                  self.ps[idx], self.n_botts[idx], self.n_surfs[idx],
                                  self.source_depths[idx])

    def __add__(self, other):
        """Concatenate along the launch-angle dimension (REF/ray_objects.py:290-345).  Unlike
        the reference (whose ``__add__`` rebuilds Rays without re-negating and so flips the
        sign of zs/ps, Q3), the stored convention is preserved."""
        if not isinstance(other, RayFan):
            raise TypeError("Can only add RayFan objects together")
        if not np.array_equal(self.rs[0], other.rs[0]):
            raise ValueError("Range arrays (rs) must be equivalent for concatenation")
        cat = np.concatenate
        return RayFan.from_arrays(cat([self.thetas, other.thetas]), cat([self.rs, other.rs]),
                                  cat([self.ts, other.ts]), cat([self.zs, other.zs]),
                                  cat([self.ps, other.ps]), cat([self.n_botts, other.n_botts]),
                                  cat([self.n_surfs, other.n_surfs]),
                                  cat([self.source_depths, other.source_depths]))

    def save_mat(self, filename):
        """.mat export with the reference's schema (REF/ray_objects.py:262-288)."""
        from scipy import io
        io.savemat(filename, {"rayfan": {
            "thetas": self.thetas, "xs": self.rs, "ts": self.ts, "zs": self.zs, "ps": self.ps,
            "n_botts": self.n_botts, "n_surfs": self.n_surfs, "source_depths": self.source_depths}})

    # ---- plots (REF/ray_objects.py:157-260) ----
    def plot_time_front(self, include_lines=False, range_idx=-1, add_colorbar=True, ray_id=False,
                        **kwargs):
        from matplotlib import pyplot as plt
        if include_lines:
            plt.plot(self.ts[:, range_idx], self.zs[:, range_idx], c="#aaaaaa", lw=0.5, zorder=5)
        kw = {"c": self.thetas, "cmap": "viridis", "s": 2, "lw": 0, "zorder": 6}
        kw.update(kwargs)
        if ray_id:
            cats = np.unique(self.ray_ids)
            colors = plt.cm.tab20(np.linspace(0, 1, len(cats)))
            lut = dict(zip(cats, colors))
            kw.update({"c": [lut[c] for c in self.ray_ids]})
            kw.pop("cmap", None)
            add_colorbar = False
        plt.scatter(self.ts[:, range_idx], self.zs[:, range_idx], **kw)
        if add_colorbar:
            plt.colorbar(label="launch angle [°]")
        plt.xlabel("time [s]")
        plt.ylabel("depth [m]")

    def plot_ray_fan(self, **kwargs):
        from matplotlib import pyplot as plt
        a = 10 * 1 / max(len(self.thetas), 1)
        kw = {"c": "k", "lw": 1, "alpha": 1 if (a > 1 or a < 0) else a}
        kw.update(kwargs)
        plt.plot(self.rs.T, self.zs.T, **kw)

    def plot_depth_v_angle(self, include_line=False, **kwargs):
        from matplotlib import pyplot as plt
        kw = {"c": self.thetas, "cmap": "viridis", "s": 2, "lw": 0, "zorder": 6}
        kw.update(kwargs)
        if include_line:
            plt.plot(self.thetas, self.zs[:, -1], c="#aaaaaa", lw=0.5)
        plt.scatter(self.thetas, self.zs[:, -1], **kw)
        plt.xlabel("launch angle [°]")
        plt.ylabel("depth [m]")


class EigenRays:
    """Eigenrays per receiver depth (REF/ray_objects.py:433-548)."""

    def __init__(self, receiver_depths, eigenray_dict, environment, num_eigenrays,
                 num_eigenrays_found, failed_eray_theta_brackets):
        from .host_physics import ray_angle
        from .environment import OceanEnvironment2D
        self.receiver_depths = receiver_depths
        self.rs, self.ts, self.zs, self.ps = {}, {}, {}, {}
        self.received_angles, self.launch_angles = {}, {}
        self.n_botts, self.n_surfs = {}, {}
        self.ray_id, self.ray_id_int = {}, {}
        self.num_eigenrays = num_eigenrays
        self.num_eigenrays_found = num_eigenrays_found
        self.failed_eray_theta_brackets = failed_eray_theta_brackets
        cin, rin, zin = OceanEnvironment2D._range_depth(environment.sound_speed)
        for ridx in range(len(receiver_depths)):
            fan = eigenray_dict[ridx]
            if not isinstance(fan, RayFan):
                fan = RayFan(fan)
            self.rs[ridx], self.ts[ridx], self.zs[ridx], self.ps[ridx] = fan.rs, fan.ts, fan.zs, fan.ps
            self.n_botts[ridx], self.n_surfs[ridx] = fan.n_botts, fan.n_surfs
            ang, ids, ids_int = [], [], []
            for k in range(len(fan)):
                # REF/ray_objects.py:521-534: received angle from the stored (negative-z)
                # state and the non-flat-earth table, exactly as the reference does (Q13)
                y_last = np.array([fan.ts[k, -1], fan.zs[k, -1], fan.ps[k, -1]])
                with np.errstate(invalid="ignore"):
                    theta, _ = ray_angle(fan.rs[k, -1], y_last, cin, rin, zin)
                ang.append(theta)
                rid = np.sum(np.diff(np.sign(fan.ps[k, :])) != 0) * np.sign(fan.thetas[k])
                flag = "" if (fan.n_botts[k] == 0 and fan.n_surfs[k] == 0) else "b"
                ids.append(f"{rid}{flag}")
                ids_int.append(int(rid))
            self.received_angles[ridx] = np.array(ang)
            self.launch_angles[ridx] = fan.thetas
            self.ray_id[ridx] = np.array(ids)
            self.ray_id_int[ridx] = np.array(ids_int)

    def plot_angle_time(self, ridxs=None, **kwargs):
        """Received angle against arrival time (REF/ray_objects.py:550-561)."""
        from matplotlib import pyplot as plt
        for ridx in (ridxs if ridxs is not None else list(self.received_angles.keys())):
            plt.scatter(self.ts[ridx][:, -1], self.received_angles[ridx], **kwargs)
        plt.xlabel("time [s]")
        plt.ylabel("received angle [deg]")
        plt.title("Received Angle vs Time")

    def plot(self, ridxs=[0], **kwargs):
        """All eigenrays of the given receiver-depth indices (REF/ray_objects.py:563-585)."""
        from matplotlib import pyplot as plt
        if isinstance(ridxs, (int, np.integer)):
            ridxs = [ridxs]
        kw = {"c": "k"}
        kw.update(kwargs)
        for ridx in ridxs:
            plt.plot(self.rs[ridx].T, self.zs[ridx].T, **kw)
        plt.xlabel("range [m]")
        plt.ylabel("depth [m]")
        plt.title("Eigen Rays")
        if len(ridxs) and self.zs[ridxs[-1]].size:
            plt.ylim([self.zs[ridxs[-1]].min(), self.zs[ridxs[-1]].max()])

    def plot_ducted(self, **kwargs):
        """The eigenrays that never touch a boundary, all receiver depths (REF/ray_objects.py:587-602; the
        reference plots -z here, depth positive down)."""
        from matplotlib import pyplot as plt
        kw = {"c": "k"}
        kw.update(kwargs)
        for ridx in self.ray_id.keys():
            mask = (self.n_botts[ridx] == 0) & (self.n_surfs[ridx] == 0)
            plt.plot(self.rs[ridx][mask].T, -self.zs[ridx][mask].T, **kw)
        plt.xlabel("range [m]")
        plt.ylabel("depth [m]")
        plt.title("Ducted Eigen Rays")

    def save_mat(self, filename):
        """.mat export with the reference's schema (REF/ray_objects.py:604-636): one struct per receiver depth,
        ``eigenrays.receiver_depth_<k>.{receiver_depth, xs, ts, zs, ps, received_angles, launch_angles, ray_id,
        ray_id_int, n_bottom, n_surface, source_depth, num_eigenrays, num_eigenrays_found}``.  (The per-depth counts
        are dictionaries in memory; MATLAB field names must be strings, so their keys are written as text.)"""
        from scipy import io

        def _fields(d):
            return {("k_" + str(k).replace(".", "p").replace("-", "m")): v for k, v in dict(d).items()}
        data = {}
        for ridx, rdepth in enumerate(self.receiver_depths):
            data[f"receiver_depth_{ridx}"] = {
                "receiver_depth": rdepth, "xs": self.rs[ridx], "ts": self.ts[ridx], "zs": self.zs[ridx], "ps": self.ps[ridx],
                "received_angles": self.received_angles[ridx], "launch_angles": self.launch_angles[ridx],
                "ray_id": self.ray_id[ridx], "ray_id_int": self.ray_id_int[ridx],
                "n_bottom": self.n_botts[ridx] if hasattr(self, "n_botts") else np.nan,
                "n_surface": self.n_surfs[ridx] if hasattr(self, "n_surfs") else np.nan,
                "source_depth": self.source_depths[ridx] if hasattr(self, "source_depths") else np.nan,
                "num_eigenrays": _fields(self.num_eigenrays), "num_eigenrays_found": _fields(self.num_eigenrays_found)}
        io.savemat(filename, {"eigenrays": data})


__all__ = ["Ray", "RayFan", "EigenRays"]
