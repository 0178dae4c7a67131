"""NumPy band engines for the CPU/gloo tests of gcmiipy_amd.bands.BandRunner.

TEST INFRASTRUCTURE: the per-band arithmetic is the oracle's, evaluated on the band
extended by its two ghost rows on either side (np.roll's wrap then only pollutes
ghost rows, which are discarded).  The product's engine is HipBandEngine."""
import numpy as np
import torch

from oracle import sw2d, sw2d_temp, tracer, dynamics, physics, geometry as ogeo

G = 2  # ghost rows per side


class NumpyBand2D:
    """GCM_SW2D / GCM_SW2D_TEMP(+van Leer tracer) on one band; state: dict name -> (H+4, W)."""
    phases = 1

    def __init__(self, fields, dx, temp, log=None, halo_steps=1):
        self.G = G * halo_steps                 # deep halo: 2 ghost rows per local step
        self.steps_per_exchange = halo_steps
        self.f = {k: np.pad(v, ((self.G, self.G), (0, 0))) for k, v in fields.items()}
        self.names = sorted(fields)
        self.H = next(iter(fields.values())).shape[0]
        self.dx, self.temp = dx, temp
        self.rb = [None, None]
        self.log = log if log is not None else []

    def _full_step(self, dt):
        f = self.f
        if self.temp:
            u, v, p, t = sw2d_temp.matsumo_scheme(f["u"], f["v"], f["p"], f["t"], self.dx, dt)
            q = tracer.limited_advection(dt, (self.dx, self.dx), np.stack([f["v"], f["u"]]), f["q"])
            return dict(u=u, v=v, p=p, t=t, q=q)
        u, v, p = sw2d.matsumo_scheme(f["u"], f["v"], f["p"], self.dx, dt)
        return dict(u=u, v=v, p=p)

    def send_buffer(self, side):
        rows = slice(self.G, 2 * self.G) if side == 0 else slice(self.H, self.H + self.G)
        return torch.from_numpy(np.concatenate([self.f[k][rows].ravel() for k in self.names]))

    def recv_buffer(self, side):
        n = sum(self.f[k][:self.G].size for k in self.names)
        self.rb[side] = torch.empty(n, dtype=torch.float64)
        return self.rb[side]

    def unpack(self, side):
        buf = self.rb[side].numpy()
        rows = slice(0, self.G) if side == 0 else slice(self.H + self.G, self.H + 2 * self.G)
        off = 0
        for k in self.names:
            n = self.f[k][rows].size
            self.f[k][rows] = buf[off:off + n].reshape(self.f[k][rows].shape)
            off += n
        self.log.append("unpack%d" % side)

    def comm_begin(self):
        self.log.append("comm_begin")

    def comm_end(self):
        self.log.append("comm_end")

    def compute_overlapped(self, phase, dt):
        # interior rows do not depend on ghost rows: computing them BEFORE the ghosts arrive
        # must give the same numbers
        self.log.append("interior")
        self.interior = {k: v[2 * G:self.H] .copy() for k, v in self._full_step(dt).items()}

    def compute_after(self, phase, dt):
        self.log.append("boundary")
        new = self._full_step(dt)
        for k in self.names:
            if self.H > 2 * G:
                assert np.array_equal(new[k][2 * G:self.H], self.interior[k]), "interior used ghosts"
            self.f[k][G:self.H + G] = new[k][G:self.H + G]

    def step_all(self, dt):
        # deep-halo local step: the whole extended band; np.roll's wrap pollutes two more ghost
        # rows per step, never the interior within `halo_steps` steps
        assert self.steps_per_exchange > 1
        self.f.update(self._full_step(dt))

    def interior_state(self):
        return {k: v[self.G:self.H + self.G].copy() for k, v in self.f.items()}


def band_geom(global_geom, row0, nrows):
    """oracle Geom for rows [row0-G, row0+nrows+G) of a global geometry (tables wrap)"""
    Hg = global_geom.height
    rows = (np.arange(row0 - G, row0 + nrows + G)) % Hg
    g = ogeo.Geom(nrows + 2 * G, global_geom.width, global_geom.layers)
    for k in ("sige", "sigt", "sigb", "dsig", "sig"):
        setattr(g, k, getattr(global_geom, k))
    g.dx_j = global_geom.dx_j[:, rows, :]
    g.dx_h = global_geom.dx_h[:, rows, :]
    g.dy, g.ptop = global_geom.dy, global_geom.ptop
    g.heightmap = global_geom.heightmap[rows]
    g.lat, g.long = global_geom.lat[rows], global_geom.long        # (the column physics)
    return g


class NumpyBandPE:
    """GCM_PE25D on one band; two phases (predictor / corrector), the exchanged state is the
    current one before the predictor and the predicted one before the corrector."""
    phases = 2

    def __init__(self, p, u, v, t, q, global_geom, row0, edge_first=False, gt=None, utc=0.0):
        """gt (the band's rows of the ground temperature) switches the column physics on: after every dynamics step
        solar_timestep (no_limits_2_5d.py:66-75) on the band's own rows AND its ghost rows -- the ghost rows of theta as
        the post-corrector exchange delivered them, those of gt as they travelled with the messages; no third
        exchange per step"""
        self.edge_first = edge_first
        self.gt = None if gt is None else np.pad(gt, ((G, G), (0, 0)))
        self.utc = utc
        self.pending = None
        self.H = p.shape[0]
        self.row0, self.Hg = row0, global_geom.height
        self.geom = band_geom(global_geom, row0, self.H)
        pad2 = lambda a: np.pad(a, ((G, G), (0, 0)))
        pad3 = lambda a: np.pad(a, ((0, 0), (G, G), (0, 0)))
        self.cur = [pad2(p), pad3(u), pad3(v), pad3(t), pad3(q)]
        self.star = None
        self.rb = [None, None]

    def _xstate(self):
        if self.pending is not None:
            return self.pending
        return self.star if self.star is not None else self.cur

    # edge-first protocol (HipBandEngine.compute_edges / compute_interior): the whole stage is
    # computed in `compute_edges`; what matters here is WHICH state the runner exchanges and
    # when its ghosts are written
    def compute_edges(self, stage, dt):
        if stage == 0:
            self.star = self._half(self.cur, dt)
        else:
            self.pending = self._half(self.star, dt)

    def compute_interior(self, stage, dt):
        if stage == 1:
            for a, b in zip(self.cur, self.pending):
                if a.ndim == 2:
                    a[G:self.H + G] = b[G:self.H + G]
                else:
                    a[:, G:self.H + G] = b[:, G:self.H + G]
            self.star = None
            self.pending = None

    @staticmethod
    def _rows(a, rows):
        return a[rows] if a.ndim == 2 else a[:, rows]

    def _xarrays(self):
        return list(self._xstate()) + ([self.gt] if self.gt is not None else [])

    def send_buffer(self, side):
        rows = slice(G, 2 * G) if side == 0 else slice(self.H, self.H + G)
        return torch.from_numpy(np.concatenate([self._rows(a, rows).ravel() for a in self._xarrays()]))

    def recv_buffer(self, side):
        n = sum(self._rows(a, slice(0, G)).size for a in self._xarrays())
        self.rb[side] = torch.empty(n, dtype=torch.float64)
        return self.rb[side]

    def physics_step(self, dt):
        if self.gt is None:
            return
        t_n, gt_n = physics.solar_timestep(self.cur[3], self.cur[0], self.gt, dt, self.utc, self.geom)
        self.cur[3][...] = t_n
        self.gt[...] = gt_n
        self.utc += dt

    def unpack(self, side):
        buf = self.rb[side].numpy()
        rows = slice(0, G) if side == 0 else slice(self.H + G, self.H + 2 * G)
        off = 0
        for a in self._xarrays():
            tgt = self._rows(a, rows)
            n = tgt.size
            if a.ndim == 2:
                a[rows] = buf[off:off + n].reshape(tgt.shape)
            else:
                a[:, rows] = buf[off:off + n].reshape(tgt.shape)
            off += n

    def comm_begin(self):
        pass

    def comm_end(self):
        pass

    def compute_overlapped(self, phase, dt):
        pass

    def _half(self, stage, dt):
        out = [x.copy() for x in dynamics.half_timestep(*self.cur, *stage, dt, self.geom)]
        # the oracle zeroed v on the extended array's last row (a ghost row); the real pole-edge
        # row (dynamics.py:222) is the global last row, owned by the last band
        last = self.Hg - 1 - self.row0 + G
        if 0 <= last < self.H + 2 * G:
            out[2][:, last, :] *= 0
        return out

    def compute_after(self, phase, dt):
        if phase == 0:
            self.star = self._half(self.cur, dt)
        else:
            new = self._half(self.star, dt)
            for a, b in zip(self.cur, new):
                if a.ndim == 2:
                    a[G:self.H + G] = b[G:self.H + G]
                else:
                    a[:, G:self.H + G] = b[:, G:self.H + G]
            self.star = None

    def step_all(self, dt):
        raise AssertionError("not used with nranks > 1")

    def interior_state(self):
        return [self._rows(a, slice(G, self.H + G)).copy() for a in self.cur + ([self.gt] if self.gt is not None else [])]
