"""gcmiipy_amd -- MI355X-native Matsuno C-grid dynamical core for gcmiipy.

Host side is Python (as the reference is); the compute path is hand-written HIP
for gfx950 in libgcmcore.so behind the C ABI of include/gcmcore.h.  Modules named
after the reference's files expose the reference's call surface:

    gcmiipy_amd.matsuno_c_grid.matsumo_scheme(u, v, p, dx, dt) -> (u, v, p)
    gcmiipy_amd.matsumo_temp.matsumo_scheme(u, v, p, t, dx, dt) -> (u, v, p, t)
    gcmiipy_amd.dynamics.matsuno_timestep(p, u, v, t, q, dt, geom) -> (p, u, v, t, q)
    gcmiipy_amd.two_d.run_2d_with_ft(state, ft, steps) ...

Importing the package loads the library and raises ImportError if it is not
built: nothing here computes on the CPU.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from .core import Core, GcmError, device_count  # noqa: F401



def clear_cache():
    """free the device state the per-call drop-ins keep between calls"""
    from . import matsuno_c_grid, dynamics
    for cache in (matsuno_c_grid._cache, dynamics._cache):
        while cache:
            cache.popitem()[1].close()
    _lib.lib.gcm_ops_release_scratch()      # the operator entry points' device scratch of this thread


__all__ = ["Core", "GcmError", "device_count", "clear_cache"]
