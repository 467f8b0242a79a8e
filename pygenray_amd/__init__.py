"""pygenray_amd -- MI355X-native drop-in for pygenray's ray-fan hot path.

Same public surface as ``pygenray`` (REF/__init__.py:5-10 star-exports): ``OceanEnvironment2D``,
``shoot_rays``, ``shoot_ray``, ``find_eigenrays``, ``Ray``, ``RayFan``, ``EigenRays``, ``munk_ssp``,
``eflat`` ... -- with the per-ray integration running in hand-written HIP on gfx950 behind the
C ABI of ``include/pgr.h``.  There is no CPU fallback: importing works anywhere, shooting rays
needs ``libpgr_hip.so`` and a GPU.
"""
from .xr_lite import DataArray
from .environment import OceanEnvironment2D, munk_ssp, eflat, eflatinv, flat_earth_c
from .ray_objects import Ray, RayFan, EigenRays
from .launch_rays import shoot_rays, shoot_ray, _unpack_envi
from .eigenrays import find_eigenrays
from .host_physics import (derivsrd, bottom_bounce, surface_bounce, ray_bounding_box_event,
                           ray_angle, bilinear_interp, linear_interp, vertical_ray)
from . import _lib

# "reference" (default: bit-identical to the CPU oracle) or "contracted" (PGR_ARITH=contracted at import: FMA contraction
# allowed, ~10 % faster, no bit-parity claim) -- see pygenray_amd/_lib.py
ARITHMETIC = _lib.ARITH

__all__ = ["OceanEnvironment2D", "munk_ssp", "eflat", "eflatinv", "flat_earth_c", "DataArray", "Ray", "RayFan",
           "EigenRays", "shoot_rays", "shoot_ray", "find_eigenrays", "derivsrd", "bottom_bounce",
           "surface_bounce", "ray_bounding_box_event", "ray_angle", "bilinear_interp",
           "linear_interp", "vertical_ray"]
