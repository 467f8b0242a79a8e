"""A minimal labelled-array type standing in for ``xarray.DataArray``.

pygenray's ``OceanEnvironment2D`` takes ``xarray.DataArray`` inputs
(REF/environment.py:49-101).  xarray is an optional dependency here: real
``xarray.DataArray`` objects are accepted when xarray is installed, and this
class provides the few members the environment front end needs
(``values, dims, coords, sizes, ndim, isel, differentiate, transpose`` and
attribute access to coordinates) so the same user code runs without it::

    ssp = DataArray(c_2d, dims=["range", "depth"], coords={"range": r, "depth": z})
"""
import numpy as np


class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None):
        self.values = np.asarray(data)
        if dims is None:
            dims = [f"dim_{k}" for k in range(self.values.ndim)]
        if isinstance(dims, str):
            dims = [dims]
        self.dims = tuple(dims)
        if len(self.dims) != self.values.ndim:
            raise ValueError("number of dims does not match data dimensionality")
        self._coords = {}
        for k, v in (coords or {}).items():
            v = np.asarray(getattr(v, "values", v))
            if k in self.dims and v.shape != (self.values.shape[self.dims.index(k)],):
                raise ValueError(f"coordinate {k!r} has the wrong length")
            self._coords[k] = v
        self.name = name

    # --- xarray-like surface ---
    @property
    def ndim(self):
        return self.values.ndim

    @property
    def shape(self):
        return self.values.shape

    @property
    def sizes(self):
        return dict(zip(self.dims, self.values.shape))

    @property
    def coords(self):
        return {k: DataArray(v, dims=[k], coords={k: v}) if k in self.dims and v.ndim == 1
                else DataArray(v) for k, v in self._coords.items()}

    def __getattr__(self, name):
        c = self.__dict__.get("_coords", {})
        if name in c:
            v = c[name]
            return DataArray(v, dims=[name], coords={name: v}) if v.ndim == 1 else DataArray(v)
        raise AttributeError(name)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __len__(self):
        return len(self.values)

    def max(self):
        return self.values.max()

    def min(self):
        return self.values.min()

    def isel(self, indexers=None, **kw):
        idx = dict(indexers or {}, **kw)
        sl = [slice(None)] * self.ndim
        dims = list(self.dims)
        coords = dict(self._coords)
        for d, i in idx.items():
            ax = self.dims.index(d)
            sl[ax] = i
            if np.isscalar(i) or isinstance(i, (int, np.integer)):
                dims.remove(d)
                if d in coords:
                    coords[d] = coords[d][i]
            elif d in coords:
                coords[d] = coords[d][i]
        return DataArray(self.values[tuple(sl)], dims=dims,
                         coords={k: v for k, v in coords.items() if np.ndim(v) == 0 or k in dims})

    def transpose(self, *dims):
        order = [self.dims.index(d) for d in dims]
        return DataArray(np.transpose(self.values, order), dims=dims, coords=self._coords)

    def differentiate(self, coord, edge_order=1):
        # xarray.DataArray.differentiate == np.gradient along the named coordinate
        ax = self.dims.index(coord)
        g = np.gradient(self.values, self._coords[coord], axis=ax, edge_order=edge_order)
        return DataArray(g, dims=self.dims, coords=self._coords)

    def copy(self, deep=True):
        return DataArray(self.values.copy() if deep else self.values, dims=self.dims,
                         coords={k: (v.copy() if deep else v) for k, v in self._coords.items()})

    def __repr__(self):
        return f"<pygenray_amd.DataArray {dict(self.sizes)}>"


def is_dataarray(obj):
    """True for this class and for a real xarray.DataArray."""
    if isinstance(obj, DataArray):
        return True
    try:
        import xarray as xr  # optional
        return isinstance(obj, xr.DataArray)
    except Exception:
        return False


def coord_values(da, name):
    c = da.coords[name]
    return np.asarray(getattr(c, "values", c), dtype=float)
