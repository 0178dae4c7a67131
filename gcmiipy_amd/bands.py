"""Latitude-band decomposition across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  The
grid's H rows are split into contiguous bands; ranks form a ring that closes between
rank 0 and rank N-1 because the reference's np.roll along j is pole-to-pole periodic
(coordinates.py:40-45).  Each exchange sends the two rows at either edge of a band
to the two ring neighbours (point-to-point, no collective on the data path).

  2-D models (fused Matsuno kernel): ONE exchange per step.  The interior rows,
    which need no ghost data, are stepped on the compute stream while the exchange
    runs on a second stream; the four edge rows follow once the ghosts landed.
    With `halo_steps = k > 1` the band carries 2k ghost rows and exchanges only every k steps
    (deep halo: the rows still valid shrink by two per step) -- for small bands, where one
    exchange costs more than a step.
  GCM_PE25D: TWO exchanges per step -- the predicted state before the corrector and the new
    state before the next predictor (the corrector needs the
    neighbour's *predicted* rows, which cannot be recomputed from a 2-row halo).  Each Euler
    stage updates and packs the two edge rows of either side on the library's second stream
    while the interior rows run on the compute stream ("edge first", gcm_set_halo_buffers);
    the exchange is posted once both are queued and hides behind the interior rows.

`BandRunner` only orchestrates; the numerical work is behind an *engine*:
`HipBandEngine` (the product: a `Core` with nranks > 1) or, in the CPU/gloo tests,
a NumPy engine defined under tests/.
"""


def split_rows(global_h, nranks):
    """contiguous bands, remainder rows go to the first ranks -> [(row0, nrows)]"""
    base, extra = divmod(global_h, nranks)
    out, r0 = [], 0
    for r in range(nranks):
        n = base + (1 if r < extra else 0)
        out.append((r0, n))
        r0 += n
    return out


class BandRunner:
    """Steps one band; `dist` is torch.distributed (initialised) or None for 1 rank.

    engine protocol:
        phases                          number of exchange+compute phases per step (1 or 2)
        send_buffer(side) -> tensor     pack the 2 edge rows of `side` (0 north, 1 south)
        recv_buffer(side) -> tensor     where the neighbour's rows for that side land
        unpack(side)                    recv_buffer(side) -> ghost rows
        compute_overlapped(phase, dt)   work that needs no ghost rows (may be a no-op)
        compute_after(phase, dt)        the rest of the phase
        step_all(dt)                    single-band step (nranks == 1)
        comm_begin() / comm_end()       stream fencing around the exchange (GPU engines)
    optional:
        pack_both() -> (north, south) / unpack_both()    both sides with one launch each
        edge_first + compute_edges(stage, dt) / compute_interior(stage, dt)   GCM_PE25D phases
        mark_packed()                   the next comm_begin() waits for the work queued so far only
        steps_per_exchange, step_n(n, dt)               deep halo
        physics_step(dt)                the column physics after the dynamics step (BASELINE configs[4]:
                                        no_limits_2_5d.solar_timestep), own rows AND ghost rows -- the ghost
                                        rows are radiated locally from the neighbour's own inputs, so no third
                                        exchange per step is needed (gcm_set_physics); an engine whose library
                                        call does the whole run (band_run) does it inside that call
    """

    def __init__(self, engine, rank, nranks, dist=None, north=None, south=None):
        """north / south: ring neighbours, default (rank -+ 1) mod nranks (tests and tools run a band
        that is its own neighbour on a one-rank communicator)"""
        self.e, self.rank, self.n, self.dist = engine, rank, nranks, dist
        self.north = (rank - 1) % nranks if north is None else north
        self.south = (rank + 1) % nranks if south is None else south
        self.k = getattr(engine, "steps_per_exchange", 1)
        self.count = 0
        self._ops = None
        self.primed = False
        # an engine that can post the exchange itself (HipBandEngine over RCCL called directly, or the
        # loopback stand-in) steps a whole run with ONE library call
        self.native = nranks > 1 and hasattr(engine, "attach_ring") and engine.attach_ring(dist, self.north, self.south)

    def _pack(self):
        """-> (north send buffer, south send buffer), packed"""
        e = self.e
        if hasattr(e, "pack_both"):
            return e.pack_both()
        return e.send_buffer(0), e.send_buffer(1)

    def _unpack(self):
        if hasattr(self.e, "unpack_both"):
            self.e.unpack_both()
        else:
            self.e.unpack(0)
            self.e.unpack(1)

    def exchange_start(self, packed=None):
        d, e = self.dist, self.e
        sn, ss = self._pack() if packed is None else packed
        e.comm_begin()
        # order matters when both neighbours are the same peer (N == 2): sends go
        # north-edge first, receives take the south ghost first (it is the peer's
        # north edge), so the i-th send pairs with the peer's i-th receive.
        rs, rn = e.recv_buffer(1), e.recv_buffer(0)
        if self._ops is None or self._ops[0] != (id(sn), id(ss), id(rs), id(rn)):
            # the buffers of a GPU engine are persistent: build the P2P descriptors once
            self._ops = ((id(sn), id(ss), id(rs), id(rn)),
                         [d.P2POp(d.isend, sn, self.north), d.P2POp(d.isend, ss, self.south),
                          d.P2POp(d.irecv, rs, self.south), d.P2POp(d.irecv, rn, self.north)])
        return d.batch_isend_irecv(self._ops[1])

    def begin_stage(self, stage, dt):
        """edge-first Euler stage up to the posted exchange -> requests for _finish()"""
        self.e.compute_edges(stage, dt)
        packed = self._pack()
        if hasattr(self.e, "mark_packed"):
            self.e.mark_packed()              # the exchange will wait for the pack only
        # the interior rows are queued BEFORE the exchange is posted: posting it costs
        # host time, which the GPU then spends in the update kernel instead of idle
        self.e.compute_interior(stage, dt)
        return self.exchange_start(packed)

    def _finish(self, reqs):
        for r in reqs:
            r.wait()
        self.e.comm_end()
        self._unpack()

    def step(self, dt):
        if self.n == 1:
            self.e.step_all(dt)
            return
        if self.k > 1:                      # deep halo: exchange, then k purely local steps
            if self.count % self.k == 0:
                self._finish(self.exchange_start())
            self.e.step_all(dt)
            self.count += 1
            return
        if getattr(self.e, "edge_first", False):
            if not self.primed:               # ghosts of the initial state, once
                self._finish(self.exchange_start())
                self.primed = True
            for stage in range(2):
                self._finish(self.begin_stage(stage, dt))
            self._physics(dt)
            return
        for phase in range(self.e.phases):
            reqs = self.exchange_start()
            self.e.compute_overlapped(phase, dt)      # overlaps the exchange
            self._finish(reqs)
            self.e.compute_after(phase, dt)
        self._physics(dt)

    def _physics(self, dt):
        ps = getattr(self.e, "physics_step", None)
        if ps is not None:
            ps(dt)

    def run(self, nsteps, dt):
        """`nsteps` steps; with a deep halo the k local steps between two exchanges are one
        library call (no per-step host work)."""
        if self.native:
            self.e.band_run(nsteps, dt)
            return
        if self.n > 1 and self.k > 1 and hasattr(self.e, "step_n"):
            done = 0
            while done < nsteps:
                left = self.k - self.count % self.k
                if left == self.k:
                    self._finish(self.exchange_start())
                n = min(left, nsteps - done)
                self.e.step_n(n, dt)
                self.count += n
                done += n
            return
        for _ in range(nsteps):
            self.step(dt)


class LoopbackExchange:
    """Diagnostic stand-in for torch.distributed inside BandRunner: every send lands in the
    matching receive buffer of the SAME rank (a device-local copy on the comm stream).  The band
    then steps with all of its launches and stream dependencies but no xGMI traffic, which
    separates what the kernels cost from what the exchange costs (bench.py, tools_band_time.py).
    The numbers it produces are not a model state (the ghost rows are the band's own edge rows)."""

    class _Req:
        def wait(self):
            pass

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    isend, irecv = "isend", "irecv"

    def batch_isend_irecv(self, ops):
        sends = [o.tensor for o in ops if o.op == "isend"]
        recvs = [o.tensor for o in ops if o.op == "irecv"]
        for s, r in zip(sends, recvs):
            r.copy_(s, non_blocking=True)
        return [self._Req()]


class HipBandEngine:
    """A `Core` band + torch CUDA buffers/streams for the exchange."""

    def __init__(self, core, torch, overlap=True, stream_aware=True):
        """stream_aware: the backend orders its transfers after the current CUDA stream (nccl =
        RCCL does).  For a backend that does not (gloo, used by the one-GPU test) the pack
        kernels are host-synchronised before the send is posted."""
        from . import _lib
        self.stream_aware = stream_aware
        self.c, self.torch = core, torch
        self.pe = core.model == _lib.PE25D
        self.phases = 2 if self.pe else 1
        self.steps_per_exchange = getattr(core, "halo_steps", 1)
        self.edge_first = self.pe and overlap
        nbytes = core.halo_bytes()
        dev = torch.device("cuda", torch.cuda.current_device())
        mk = lambda: torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
        self.sbuf, self.rbuf = [mk(), mk()], [mk(), mk()]
        self.compute = torch.cuda.current_stream()
        # the comm stream comes from the library, which checks that it really runs beside the
        # compute stream (two HIP streams may share a hardware queue and then run in order)
        self.comm = torch.cuda.ExternalStream(core.comm_stream()) if overlap else self.compute
        self.overlap = overlap
        self._ctx = None
        self._packed = torch.cuda.Event()
        self._wait_packed = False
        # GCM_PE25D: the library updates and packs the edge rows on its second stream while the
        # interior rows run on the compute stream (gcm_set_halo_buffers / gcm_wait_edges)
        self.async_edges = self.edge_first
        self._edges_pending = False
        if self.async_edges:
            core.set_halo_buffers(self.sbuf[0].data_ptr(), self.sbuf[1].data_ptr())

    def _s(self, stream):
        return stream.cuda_stream

    def attach_ring(self, ring, north, south):
        """-> True if the library can post this ring's exchange itself (gcm_set_exchange): RCCL called
        directly (gcmiipy_amd.rccl.RcclP2P) or the loopback stand-in; GCM_BAND_HOST_LOOP=1 keeps the
        host-driven sequence (the reference path of the tests)."""
        import os
        from .rccl import RcclP2P
        if os.environ.get("GCM_BAND_HOST_LOOP") or not self.overlap:
            return False
        if isinstance(ring, RcclP2P):
            rccl = ring
        elif isinstance(ring, LoopbackExchange):
            rccl = None
        else:
            return False
        self.c.set_exchange(self.sbuf[0].data_ptr(), self.sbuf[1].data_ptr(), self.rbuf[0].data_ptr(),
                            self.rbuf[1].data_ptr(), rccl, north, south)
        return True

    def band_run(self, n, dt):
        self.c.band_run(n, dt)

    def set_physics(self, geom, utc=0.0):
        """solar_timestep after every dynamics step (no_limits_2_5d.py:66-75, t_lw = 0.1, t_sw = 0.9, albedo = 0.3):
        inside the library's gcm_band_run, or from physics_step() when the host drives the exchange"""
        self.c.set_physics(geom, utc)
        self._phys = [geom, float(utc)]

    def physics_step(self, dt):
        """host-driven band step: own rows and ghost rows by one gcm_solar_step on the compute stream, behind the
        unpack of the post-corrector exchange"""
        ph = getattr(self, "_phys", None)
        if ph is None:
            return
        self.c.solar_step(ph[0], dt, ph[1])
        ph[1] += dt

    def send_buffer(self, side):
        self.c.halo_pack(side, self.sbuf[side].data_ptr(), self._s(self.compute))
        if not self.stream_aware:
            self.compute.synchronize()
        return self.sbuf[side]

    def pack_both(self):
        if self._edges_pending:             # packed by the library inside compute_edges()
            return self.sbuf[0], self.sbuf[1]
        self.c.halo_pack2(self.sbuf[0].data_ptr(), self.sbuf[1].data_ptr(), self._s(self.compute))
        if not self.stream_aware:
            self.compute.synchronize()
        return self.sbuf[0], self.sbuf[1]

    def unpack_both(self):
        self.c.halo_unpack2(self.rbuf[0].data_ptr(), self.rbuf[1].data_ptr(), self._s(self.compute))

    def mark_packed(self):
        if self.overlap and not self._edges_pending:
            self._packed.record(self.compute)
            self._wait_packed = True

    def recv_buffer(self, side):
        return self.rbuf[side]

    def comm_begin(self):
        # the collective library orders its work after the *current* stream: make that
        # the comm stream, which waits for the pack kernels, so the interior step that
        # is launched next on the compute stream runs concurrently with the exchange
        if self._edges_pending:             # the send waits for the library's edge-row pack only
            self._edges_pending = False
            self.c.wait_edges(self._s(self.comm))
            if not self.stream_aware:
                self.comm.synchronize()
            if self.overlap:
                self._ctx = self.torch.cuda.stream(self.comm)
                self._ctx.__enter__()
            return
        if not self.overlap:
            return
        if self._wait_packed:               # work queued on the compute stream after the pack
            self.comm.wait_event(self._packed)      # (the interior rows) is not waited for
            self._wait_packed = False
        else:
            self.comm.wait_stream(self.compute)
        self._ctx = self.torch.cuda.stream(self.comm)
        self._ctx.__enter__()

    def _leave_comm(self):
        if self._ctx is not None:
            self._ctx.__exit__(None, None, None)
            self._ctx = None

    def comm_end(self):
        self._leave_comm()
        if self.overlap:
            self.compute.wait_stream(self.comm)

    def unpack(self, side):
        self.c.halo_unpack(side, self.rbuf[side].data_ptr(), self._s(self.compute))

    def compute_overlapped(self, phase, dt):
        self._leave_comm()                  # compute work goes to the compute stream
        if not self.pe:
            self.c.step_interior(dt, self._s(self.compute))

    def compute_after(self, phase, dt):
        if self.pe:                         # phase 0: predictor, phase 1: corrector + swap
            if phase == 0:
                self.c.step_interior(dt, self._s(self.compute))
            else:
                self.c.step_boundary(dt, self._s(self.compute))
        else:
            self.c.step_boundary(dt, self._s(self.compute))

    # GCM_PE25D edge-first protocol (gcm_step_phase)
    def compute_edges(self, stage, dt):
        self.c.step_phase(2 * stage, dt, self._s(self.compute))
        self._edges_pending = self.async_edges

    def compute_interior(self, stage, dt):
        self._leave_comm()
        self.c.step_phase(2 * stage + 1, dt, self._s(self.compute))

    def step_all(self, dt):
        self.c.step(1, dt)

    def step_n(self, n, dt):
        self.c.step(n, dt)
