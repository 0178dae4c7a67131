"""The 2-D operators the step kernels are fused from, callable one by one (`gcm_sw2d_op`): the
functions that matsuno_c_grid.py, viscosity.py, matsumo_temp.py and temperature.py export.  The
drop-in modules of those names re-export them; this module holds the marshalling."""
import numpy as np

from . import _lib
from ._lib import lib
from .core import as_f64
from .two_d import _ops_check, _p
from .units import strip, scalar


def stencil(kind, arrays, dx=1.0, mu=0.0):
    """a stencil operator on 2-D fields (periodic in both axes)"""
    a = [as_f64(strip(x)[0], name="field") for x in arrays]
    if a[0].ndim != 2:
        raise ValueError("the field must be 2-D")
    a = [as_f64(x, a[0].shape, "field") for x in a] + [None] * (3 - len(a))
    out = np.empty(a[0].shape)
    _ops_check(lib.gcm_sw2d_op(kind, a[0].shape[1], a[0].shape[0], scalar(dx), scalar(mu), _p(a[0]), _p(a[1]),
                               _p(a[2]), _p(out)))
    return out


def elementwise(kind, x0, x1, dx=1.0):
    """an elementwise operator on arrays (or scalars) of one shape (temperature.py:9,17 assert it)"""
    a0 = np.asarray(strip(x0)[0], dtype=np.float64)
    a1 = np.asarray(strip(x1)[0], dtype=np.float64)
    if a0.shape != a1.shape:
        raise ValueError("shapes differ: %s vs %s" % (a0.shape, a1.shape))
    f0, f1 = np.ascontiguousarray(a0).reshape(-1), np.ascontiguousarray(a1).reshape(-1)
    out = np.empty(f0.shape)
    if f0.size:
        _ops_check(lib.gcm_sw2d_op(kind, f0.size, 1, scalar(dx), 0.0, _p(f0), _p(f1), None, _p(out)))
    out = out.reshape(a0.shape)
    return out if a0.ndim else float(out)


# matsuno_c_grid.py
def advection_of_velocity_u(u, v, dx): return stencil(_lib.OP_ADV_U, (u, v), dx)              # :15-51
def advection_of_velocity_v(u, v, dx): return stencil(_lib.OP_ADV_V, (u, v), dx)              # :54-80
def geopotential_gradient_u(p, dx): return stencil(_lib.OP_GEO_GRAD_U, (p,), dx)              # :97-100
def geopotential_gradient_v(p, dx): return stencil(_lib.OP_GEO_GRAD_V, (p,), dx)              # :103-106
def advection_of_geopotential(u, v, p, dx): return stencil(_lib.OP_ADV_GEO, (u, v, p), dx)    # :109-118


# viscosity.py
def finite_laplacian_2d(q, dx): return stencil(_lib.OP_LAPLACIAN, (q,), dx)                    # :12-19
def incompressible_viscosity_2d(u, mu, dx): return stencil(_lib.OP_VISCOSITY, (u,), dx, mu)    # :22-25


# matsumo_temp.py
def density_from(p, t): return elementwise(_lib.OP_DENSITY_FROM, p, t)                         # :13-19
def potential_temperature(p, t): return elementwise(_lib.OP_TO_POTENTIAL_TEMP, t, p)           # :22-25
def scaling(pa, t, dx): return elementwise(_lib.OP_SCALING, pa, t, dx)                         # :28-30
def unscaling(pb, tt, dx): return elementwise(_lib.OP_UNSCALING, pb, tt, dx)                   # :33-35
def geopotential_from(rho, p): return elementwise(_lib.OP_GEOPOTENTIAL_FROM, rho, p)           # :45-47


def advect_t(t, u, v, pa, pb, dx, dt):                                                        # :38-42
    scaled_t = scaling(pa, t, dx)
    tt = scaled_t - scalar(dt) * advection_of_geopotential(u, v, scaled_t, dx)
    return unscaling(pb, tt, dx)


# temperature.py
def to_true_temp(t, p): return elementwise(_lib.OP_TO_TRUE_TEMP, t, p)                         # :7-12
def to_potential_temp(tt, p): return elementwise(_lib.OP_TO_POTENTIAL_TEMP, tt, p)             # :15-19
def to_density(tt, p): return elementwise(_lib.OP_TO_DENSITY, tt, p)                           # :22-24
