"""`Core`: one resident model state on one MI355X behind the C ABI.

This is the fast path: state stays in HBM, `step(n)` launches n fused Matsuno
steps, `get_state()` copies back once.  The reference-shaped drop-in functions
(matsuno_c_grid.py, matsumo_temp.py, dynamics.py ... in this package) are thin
wrappers that move arrays through a cached Core per call.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib


class GcmError(RuntimeError):
    pass


def _check(rc, h=None):
    if rc == _lib.OK:
        return
    msg = lib.gcm_last_error(h).decode() if (h or rc) else ""
    if rc == _lib.ERR_ARG:
        raise ValueError(msg or "gcmcore: bad argument")
    raise GcmError("gcmcore error %d: %s" % (rc, msg))


def as_f64(x, shape=None, name="array"):
    """float64 C-contiguous ndarray of `shape`; mirrors the reference's shape asserts
    (temperature.py:9,17; dynamics.py:203) with ValueError."""
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _tab(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class Core:
    """Owns a gcm_handle.  Field order of set/get is (p, u, v, t, q)."""

    def __init__(self, model, width, height, layers=1, dx=0.0, tracer=_lib.TRACER_NONE,
                 variant=_lib.VARIANT_AUTO, geom=None, filter=True, nranks=1, rank=0,
                 global_height=None, row0=0, device=-1, stream=None, halo_steps=1, coriolis=False, dtype="f64"):
        self.model, self.W, self.H, self.L = model, int(width), int(height), int(layers)
        self.nranks, self.rank = nranks, rank
        # what a checkpoint needs to rebuild this handle (checkpoint.save / restore)
        self.options = dict(dx=float(dx), tracer=int(tracer), variant=int(variant), filter=bool(filter),
                            nranks=int(nranks), rank=int(rank), row0=int(row0), halo_steps=int(halo_steps),
                            coriolis=bool(coriolis), dtype=dtype,
                            global_height=int(height if global_height is None else global_height))
        self.has_ground = False
        cfg = _lib.Config()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.model = model
        cfg.width, cfg.height, cfg.layers = self.W, self.H, self.L
        cfg.tracer, cfg.variant, cfg.filter = tracer, variant, 1 if filter else 0
        cfg.nranks, cfg.rank = nranks, rank
        cfg.global_height = self.H if global_height is None else int(global_height)
        self.global_height = cfg.global_height
        cfg.row0 = row0
        cfg.device = device
        cfg.halo_steps = halo_steps
        cfg.dtype = {"f64": _lib.F64, "f32": _lib.F32}[dtype]
        self.dtype = dtype
        self.halo_steps = halo_steps
        cfg.dx = float(dx)
        cfg.stream = stream
        self._keep = []
        if model == _lib.PE25D:
            if geom is None:
                raise ValueError("GCM_PE25D needs a geometry (gcmiipy_amd.geometry.gen_geometry)")
            gh = cfg.global_height

            def tab(x, n):
                a = as_f64(np.asarray(x, dtype=np.float64).reshape(-1), (n,), "geometry table")
                self._keep.append(a)
                return _tab(a)

            cfg.dy = float(geom.dy)
            cfg.ptop = float(geom.ptop)
            cfg.dx_j, cfg.dx_h = tab(geom.dx_j, gh), tab(geom.dx_h, gh)
            cfg.sig, cfg.dsig = tab(geom.sig, self.L), tab(geom.dsig, self.L)
            cfg.sigb, cfg.sigt = tab(geom.sigb, self.L), tab(geom.sigt, self.L)
            cfg.heightmap = tab(geom.heightmap, gh * self.W)
            if coriolis:
                from .geometry import coriolis_tables
                cu, cv = coriolis_tables(geom)
                cfg.cor_u, cfg.cor_v = tab(cu, gh), tab(cv, gh)
        self._h = _lib._H()
        rc = lib.gcm_create(C.byref(cfg), C.byref(self._h))
        if rc != _lib.OK:
            msg = lib.gcm_last_error(None).decode()
            self._h = None
            if rc == _lib.ERR_ARG:
                raise ValueError(msg)
            raise GcmError("gcm_create failed (%d): %s" % (rc, msg))
        self.is3d = model == _lib.PE25D
        self.fields = {_lib.SW2D: (_lib.P, _lib.U, _lib.V),
                       _lib.SW2D_TEMP: (_lib.P, _lib.U, _lib.V, _lib.T) +
                       ((_lib.Q,) if tracer != _lib.TRACER_NONE else ()),
                       }.get(model, (_lib.P, _lib.U, _lib.V, _lib.T, _lib.Q))

    # -- shapes ----------------------------------------------------------------
    def shape_of(self, field):
        if self.is3d and field != _lib.P:
            return (self.L, self.H, self.W)
        return (self.H, self.W)

    def _prep_in(self, arrs):
        out = []
        for f, a in enumerate(arrs):
            out.append(None if a is None else as_f64(a, self.shape_of(f), "puvtq"[f]))
        return out

    # -- state -------------------------------------------------------------------
    def set_state(self, p=None, u=None, v=None, t=None, q=None):
        a = self._prep_in((p, u, v, t, q))
        _check(lib.gcm_set_state(self._h, *[_ptr(x) for x in a]), self._h)

    def set_star(self, p=None, u=None, v=None, t=None, q=None):
        a = self._prep_in((p, u, v, t, q))
        _check(lib.gcm_set_star(self._h, *[_ptr(x) for x in a]), self._h)

    def _get(self, fn, fields):
        out = [np.empty(self.shape_of(f)) if f in fields else None for f in range(5)]
        _check(fn(self._h, *[_ptr(x) for x in out]), self._h)
        return out

    def get_state(self, fields=None):
        """-> [p, u, v, t, q] (None for fields not requested / not in the model)"""
        return self._get(lib.gcm_get_state, self.fields if fields is None else fields)

    def get_star(self, fields=(_lib.P, _lib.U, _lib.V, _lib.T)):
        return self._get(lib.gcm_get_star, fields)

    # -- stepping ------------------------------------------------------------------
    def step(self, nsteps, dt):
        _check(lib.gcm_step(self._h, int(nsteps), float(dt)), self._h)

    def half_step(self, stage, dt):
        _check(lib.gcm_half_step(self._h, int(stage), float(dt)), self._h)

    def get_intermediate(self, kind):
        """parity tap (GCM_PE25D): spu, pit, p_n, phi or pgfu of the last half step, as the stage kernels
        left them in the handle (gcm_get_intermediate; _lib.INT_*)"""
        two_d = kind in (_lib.INT_PIT, _lib.INT_PN)
        out = np.empty((self.H, self.W) if two_d else (self.L, self.H, self.W))
        _check(lib.gcm_get_intermediate(self._h, int(kind), _ptr(out)), self._h)
        return out

    def snapshot(self):
        """device-side copy of the current state (2-D models)"""
        _check(lib.gcm_snapshot(self._h), self._h)

    def restore(self):
        """current state <- snapshot, asynchronously on the handle's stream"""
        _check(lib.gcm_restore(self._h), self._h)

    def sync(self):
        _check(lib.gcm_sync(self._h), self._h)

    def diag(self, kind):
        out = C.c_double()
        _check(lib.gcm_diag(self._h, kind, C.byref(out)), self._h)
        return out.value

    def energy(self, area):
        """(ke, ate, geo, total), no_limits_2_5d.calc_energy"""
        a = as_f64(np.asarray(area, dtype=np.float64).reshape(-1), name="area")
        out = (C.c_double * 4)()
        _check(lib.gcm_energy(self._h, _tab(a), a.size, out), self._h)
        return tuple(out)

    def stats(self, area):
        """the STATS record of full_timestep (no_limits_2_5d.py:85-91) by one launch and one
        synchronisation -> dict(u_max, u_min, v_max, v_min, ke=(ke, ate, geo, total), nans)"""
        a = as_f64(np.asarray(area, dtype=np.float64).reshape(-1), name="area")
        out = (C.c_double * 9)()
        _check(lib.gcm_stats(self._h, _tab(a), a.size, out), self._h)
        return {"u_max": out[0], "u_min": out[1], "v_max": out[2], "v_min": out[3],
                "ke": (out[4], out[5], out[6], out[7]), "nans": out[8]}

    def total_variation(self, field):
        """constants.get_total_variation of one field of the resident state (constants.py:105-108):
        sum |q - roll(q, -1, 0)| over axis 0 of the reference layout, by a device reduction"""
        return self.diag(_lib.DIAG_TV_P + int(field))

    def polar_filter(self, q):
        """low_pass.arakawa_1977 on a (H, W) or (n, H, W) field of this handle's grid (any n: the
        levels go through the handle `layers` at a time)"""
        a = as_f64(q, name="q")
        if a.ndim not in (2, 3) or a.shape[-2:] != (self.H, self.W):
            raise ValueError("q has shape %s, expected (..., %d, %d)" % (a.shape, self.H, self.W))
        a3 = a.reshape(-1, self.H, self.W)
        out = np.empty_like(a3)
        for k0 in range(0, a3.shape[0], self.L):
            k1 = min(k0 + self.L, a3.shape[0])
            _check(lib.gcm_polar_filter(self._h, k1 - k0, _ptr(a3[k0:k1]), _ptr(out[k0:k1])), self._h)
        return out.reshape(a.shape)

    # -- column physics (GCM_PE25D) ----------------------------------------------------
    def set_ground(self, gt):
        _check(lib.gcm_set_ground(self._h, _ptr(as_f64(gt, (self.H, self.W), "gt"))), self._h)
        self.has_ground = True

    def get_ground(self):
        out = np.empty((self.H, self.W))
        _check(lib.gcm_get_ground(self._h, _ptr(out)), self._h)
        return out

    def _latlon(self, geom):
        lat = as_f64(np.asarray(geom.lat, dtype=np.float64).reshape(-1), (self.global_height,), "lat")
        lon = as_f64(np.asarray(geom.long, dtype=np.float64).reshape(-1), (self.W,), "long")
        return lat, lon

    def grey_radiation(self, geom, utc, t_lw=0.1, t_sw=0.9, albedo=0.3):
        """-> (dTdt [L,H,W], dt_ground [H,W]), grey_solar.basic_grey_radiation"""
        lat, lon = self._latlon(geom)
        dT, dg = np.empty((self.L, self.H, self.W)), np.empty((self.H, self.W))
        _check(lib.gcm_grey_radiation(self._h, float(utc), t_lw, t_sw, albedo, _tab(lat), _tab(lon),
                                      _ptr(dT), _ptr(dg)), self._h)
        return dT, dg

    def solar_step(self, geom, dt, utc, t_lw=0.1, t_sw=0.9, albedo=0.3):
        lat, lon = self._latlon(geom)
        _check(lib.gcm_solar_step(self._h, float(dt), float(utc), t_lw, t_sw, albedo, _tab(lat),
                                  _tab(lon)), self._h)

    def set_physics(self, geom, utc=0.0, t_lw=0.1, t_sw=0.9, albedo=0.3):
        """every step of step() / band_run() from now on = the dynamics step followed by
        no_limits_2_5d.solar_timestep at the handle's clock, which then advances by dt (run_model's loop,
        no_limits_2_5d.py:229-234); geom=None switches the physics off (gcm_set_physics)"""
        if geom is None:
            _check(lib.gcm_set_physics(self._h, None), self._h)
            return
        lat, lon = self._latlon(geom)
        ph = _lib.Physics(float(utc), t_lw, t_sw, albedo, _tab(lat), _tab(lon))
        _check(lib.gcm_set_physics(self._h, C.byref(ph)), self._h)

    def utc(self):
        out = C.c_double()
        _check(lib.gcm_get_utc(self._h, C.byref(out)), self._h)
        return out.value

    def time_steps(self, nsteps, dt, per_kernel=True):
        ms, kms = C.c_double(), C.c_double()
        _check(lib.gcm_time_steps(self._h, int(nsteps), float(dt), C.byref(ms),
                                  C.byref(kms) if per_kernel else None), self._h)
        return ms.value, (kms.value if per_kernel else None)

    # -- latitude-band plumbing ------------------------------------------------------
    def halo_bytes(self):
        return lib.gcm_halo_bytes(self._h)

    def halo_pack(self, side, dev_ptr, stream=None):
        _check(lib.gcm_halo_pack(self._h, side, dev_ptr, stream), self._h)

    def halo_unpack(self, side, dev_ptr, stream=None):
        _check(lib.gcm_halo_unpack(self._h, side, dev_ptr, stream), self._h)

    def halo_pack2(self, north_ptr, south_ptr, stream=None):
        _check(lib.gcm_halo_pack2(self._h, north_ptr, south_ptr, stream), self._h)

    def halo_unpack2(self, north_ptr, south_ptr, stream=None):
        _check(lib.gcm_halo_unpack2(self._h, north_ptr, south_ptr, stream), self._h)

    def set_halo_buffers(self, north_ptr, south_ptr):
        _check(lib.gcm_set_halo_buffers(self._h, north_ptr, south_ptr), self._h)

    def wait_edges(self, stream):
        _check(lib.gcm_wait_edges(self._h, stream), self._h)

    def comm_stream(self):
        """hipStream_t (int) of a handle-owned stream measured to run beside the compute stream"""
        out = C.c_void_p()
        _check(lib.gcm_comm_stream(self._h, C.byref(out)), self._h)
        return out.value

    def set_exchange(self, send_north, send_south, recv_north, recv_south, rccl=None, north=0, south=0):
        """register the ghost-row exchange the library posts itself (gcm_set_exchange).  `rccl`: a
        gcmiipy_amd.rccl.RcclP2P (its communicator and the addresses of its librccl entry points);
        None: loopback, the band is its own neighbour (device-local copies)"""
        x = _lib.Exchange()
        x.north, x.south = north, south
        x.send_north, x.send_south, x.recv_north, x.recv_south = send_north, send_south, recv_north, recv_south
        if rccl is not None:
            x.comm = rccl.comm
            x.send, x.recv, x.group_start, x.group_end = rccl.entry_points()
        _check(lib.gcm_set_exchange(self._h, C.byref(x)), self._h)

    def set_band_overlap(self, on):
        """deep-halo 2-D bands: hide the exchange behind interior rows (gcm_set_band_overlap)"""
        _check(lib.gcm_set_band_overlap(self._h, 1 if on else 0), self._h)

    def band_run(self, nsteps, dt):
        """`nsteps` full band steps, exchanges included, one library call (gcm_band_run)"""
        _check(lib.gcm_band_run(self._h, int(nsteps), float(dt)), self._h)

    def step_interior(self, dt, stream=None):
        _check(lib.gcm_step_interior(self._h, float(dt), stream), self._h)

    def step_boundary(self, dt, stream=None):
        _check(lib.gcm_step_boundary(self._h, float(dt), stream), self._h)

    def step_phase(self, phase, dt, stream=None):
        _check(lib.gcm_step_phase(self._h, int(phase), float(dt), stream), self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib.gcm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    return lib.gcm_device_count()
