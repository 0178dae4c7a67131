"""Drop-in for the reference's dynamics.py call surface (2.5-D sigma-level primitive
equations): `matsuno_timestep` and `half_timestep`, computed by the HIP kernels."""
import numpy as np

from . import _lib
from .core import Core, as_f64
from .units import strip, scalar, attach

_cache = {}


def _geom_key(geom):
    return (geom.height, geom.width, geom.layers, float(geom.dy), float(geom.ptop),
            np.asarray(geom.dx_j).tobytes(), np.asarray(geom.sig).tobytes(),
            np.asarray(geom.heightmap).tobytes())


def core_for(geom, filter=True, coriolis=False):
    key = (_geom_key(geom), filter, coriolis)
    c = _cache.get(key)
    if c is None:
        if len(_cache) > 4:
            _cache.popitem()[1].close()
        c = _cache[key] = Core(_lib.PE25D, geom.width, geom.height, geom.layers, geom=geom,
                               filter=filter, coriolis=coriolis)
    return c


def _prep(p, u, v, t, q, geom):
    shp3, shp2 = (geom.layers, geom.height, geom.width), (geom.height, geom.width)
    vals, units = zip(*(strip(x) for x in (p, u, v, t, q)))
    arrs = [as_f64(vals[0], shp2, "p")] + [as_f64(a, shp3, n) for a, n in zip(vals[1:], "uvtq")]
    return arrs, units


def _wrap_out(arrs, units):
    return tuple(attach(a, un) for a, un in zip(arrs, units))


def half_timestep(p, u, v, t, q, sp, su, sv, st, sq, dt, geom):
    """dynamics.py:183-227: one Euler stage from base (p,u,v,t,q) with tendencies evaluated
    on the stage state; returns fresh (p_n, u_n, v_n, t_n, q_n)."""
    base, units = _prep(p, u, v, t, q, geom)
    stage, _ = _prep(sp, su, sv, st, sq, geom)
    c = core_for(geom)
    c.set_state(*base)
    c.set_star(*stage)
    c.half_step(1, scalar(dt))          # corrector form: base + stage -> new current state
    return _wrap_out(c.get_state(), units)


def matsuno_timestep(p, u, v, t, q, dt, geom, boundary_conditions=None, coriolis=False):
    """dynamics.py:230-237.  `coriolis=True` switches on the Coriolis terms the reference keeps
    behind `if False` (dynamics.py:82-92).  With a Python `boundary_conditions(sp,su,sv,st,sq,dt,geom)` hook
    the predicted state makes a host round trip between the stages (documented slow path);
    with None both stages stay on the device."""
    base, units = _prep(p, u, v, t, q, geom)
    c = core_for(geom, coriolis=coriolis)
    c.set_state(*base)
    dts = scalar(dt)
    if boundary_conditions is None:
        c.step(1, dts)
        return _wrap_out(c.get_state(), units)
    c.half_step(0, dts)
    star = _wrap_out(c.get_star((0, 1, 2, 3, 4)), units)
    star = boundary_conditions(*star, dt, geom)
    c.set_star(*_prep(*star, geom)[0])
    c.half_step(1, dts)
    out = _wrap_out(c.get_state(), units)
    return boundary_conditions(*out, dt, geom)


def run(p, u, v, t, q, dt, geom, steps, callback=None, every=1):
    """Device-resident loop: `steps` Matsuno steps with the state in HBM throughout;
    optional callback(p,u,v,t,q) every `every` steps (no_limits_2_5d.py:230-234)."""
    base, units = _prep(p, u, v, t, q, geom)
    c = Core(_lib.PE25D, geom.width, geom.height, geom.layers, geom=geom)
    try:
        c.set_state(*base)
        done = 0
        while done < steps:
            n = min(every, steps - done) if callback else steps - done
            c.step(n, scalar(dt))
            done += n
            if callback:
                callback(*_wrap_out(c.get_state(), units))
        return _wrap_out(c.get_state(), units)
    finally:
        c.close()


# ---- the operators the step is made of, one by one (dynamics.py:15-181), behind gcm_pe25d_op:
# arrays in the reference's layout ([k, j, i]; p: [j, i]), SI magnitudes out
def _op(kind, geom, ins, out_shapes):
    import ctypes as C
    from ._lib import lib, PeGeom
    from .core import GcmError
    L, H, W = geom.layers, geom.height, geom.width
    shp3, shp2 = (L, H, W), (H, W)
    arrs = [as_f64(strip(x)[0], shp2 if two_d else shp3, "operand") for x, two_d in ins]
    tabs = [as_f64(np.asarray(strip(getattr(geom, k))[0], dtype=np.float64).reshape(-1), (n,), k)
            for k, n in (("dx_j", H), ("dx_h", H), ("dsig", L), ("sig", L), ("sigb", L), ("sigt", L))]
    hm = as_f64(strip(geom.heightmap)[0], shp2, "heightmap")
    g = PeGeom(*[t.ctypes.data for t in tabs], hm.ctypes.data, scalar(geom.dy), scalar(geom.ptop))
    outs = [np.empty(shp2 if two_d else shp3) for two_d in out_shapes]
    pin = (C.c_void_p * 5)(*([a.ctypes.data for a in arrs] + [None] * (5 - len(arrs))))
    pout = (C.c_void_p * 4)(*([o.ctypes.data for o in outs] + [None] * (4 - len(outs))))
    rc = lib.gcm_pe25d_op(kind, W, H, L, C.addressof(g), C.byref(pin), C.byref(pout))
    if rc != _lib.OK:
        msg = lib.gcm_pe25d_op_last_error().decode()
        if rc == _lib.ERR_ARG:
            raise ValueError(msg)
        raise GcmError("gcmcore error %d: %s" % (rc, msg))
    return outs[0] if len(outs) == 1 else tuple(outs)


def calc_pu(p, u, geom=None): return _op(_lib.PEOP_CALC_PU, _need(geom, u, p), ((p, True), (u, False)), (False,))    # :15-17
def calc_pv(p, v, geom=None): return _op(_lib.PEOP_CALC_PV, _need(geom, v, p), ((p, True), (v, False)), (False,))    # :20-22
def un_pu(pu, p, geom=None): return _op(_lib.PEOP_UN_PU, _need(geom, pu, p), ((pu, False), (p, True)), (False,))     # :25-27
def un_pv(pv, p, geom=None): return _op(_lib.PEOP_UN_PV, _need(geom, pv, p), ((pv, False), (p, True)), (False,))     # :30-32
def aflux(pu, pv, geom): return _op(_lib.PEOP_AFLUX, geom, ((pu, False), (pv, False)), (True, False))              # :35-46
def advec_sig(sd, q, geom): return _op(_lib.PEOP_ADVEC_SIG, geom, ((sd, False), (q, False)), (False,))             # :49-52


def advec_m_pu(p, u, v, pu, pv, geom):                                                                             # :55-108
    return _op(_lib.PEOP_ADVEC_M_PU, geom, ((p, True), (u, False), (v, False), (pu, False), (pv, False)), (False, False))


def compute_geopotential(p, t, geom):                                                                              # :111-143
    return _op(_lib.PEOP_GEOPOTENTIAL, geom, ((p, True), (t, False)), (False,))


def pgf(p, t, geom):                                                                                               # :147-171
    return _op(_lib.PEOP_PGF, geom, ((p, True), (t, False)), (False, False, False, False))


def advec_t(pu, pv, t, geom): return _op(_lib.PEOP_ADVEC_T, geom, ((pu, False), (pv, False), (t, False)), (False,))  # :174-181


class _Shape:
    """calc_pu / calc_pv / un_pu / un_pv take no geometry in the reference: the grid is the arrays' shape"""
    def __init__(self, L, H, W):
        self.layers, self.height, self.width = L, H, W
        self.dx_j = self.dx_h = np.ones(H)
        self.dsig = self.sig = self.sigb = self.sigt = np.ones(L)
        self.heightmap = np.zeros((H, W))
        self.dy, self.ptop = 1.0, 0.0


def _need(geom, a3, a2):
    if geom is not None:
        return geom
    s = np.shape(strip(a3)[0])
    if len(s) != 3:
        raise ValueError("expected a [k, j, i] array")
    return _Shape(*s)
