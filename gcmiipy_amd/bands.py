"""Latitude-band decomposition across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  The
grid's H rows are split into contiguous bands; every step each rank sends the two
rows at either edge of its band to its ring neighbours -- the ring closes between
rank 0 and rank N-1 because the reference's np.roll along j is pole-to-pole
periodic (coordinates.py:40-45) -- and meanwhile steps the interior rows that need
no ghost data on the compute stream; the edge rows follow once the ghosts landed.

`BandRunner` only orchestrates; the numerical work is behind an *engine*:
`HipBandEngine` (the product: a `Core` with nranks > 1) or, in the CPU/gloo tests,
a NumPy engine defined under tests/.
"""
import numpy as np


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
        send_buffer(side) -> tensor      pack the 2 edge rows of `side` (0 north, 1 south)
        recv_buffer(side) -> tensor      where the neighbour's rows for that side land
        unpack(side)                     recv_buffer(side) -> ghost rows
        step_interior(dt), step_boundary(dt), step_all(dt)
        comm_begin() / comm_end()        stream fencing around the exchange (GPU engines)
    """

    def __init__(self, engine, rank, nranks, dist=None):
        self.e, self.rank, self.n, self.dist = engine, rank, nranks, dist
        self.north = (rank - 1) % nranks
        self.south = (rank + 1) % nranks

    def exchange_start(self):
        d, e = self.dist, self.e
        sn, ss = e.send_buffer(0), e.send_buffer(1)
        e.comm_begin()
        # order matters when both neighbours are the same peer (N == 2): sends go
        # north-edge first, receives take the south ghost first (it is the peer's
        # north edge), so the i-th send pairs with the peer's i-th receive.
        ops = [d.P2POp(d.isend, sn, self.north), d.P2POp(d.isend, ss, self.south),
               d.P2POp(d.irecv, e.recv_buffer(1), self.south),
               d.P2POp(d.irecv, e.recv_buffer(0), self.north)]
        return d.batch_isend_irecv(ops)

    def step(self, dt):
        if self.n == 1:
            self.e.step_all(dt)
            return
        reqs = self.exchange_start()
        self.e.step_interior(dt)          # overlaps the exchange
        for r in reqs:
            r.wait()
        self.e.comm_end()
        self.e.unpack(0)
        self.e.unpack(1)
        self.e.step_boundary(dt)


class HipBandEngine:
    """A `Core` band + torch CUDA buffers/streams for the exchange."""

    def __init__(self, core, torch):
        self.c, self.torch = core, torch
        nbytes = core.halo_bytes()
        dev = torch.device("cuda", torch.cuda.current_device())
        mk = lambda: torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
        self.sbuf, self.rbuf = [mk(), mk()], [mk(), mk()]
        self.compute = torch.cuda.current_stream()
        self.comm = torch.cuda.Stream()
        self._ctx = None

    def _s(self, stream):
        return stream.cuda_stream

    def send_buffer(self, side):
        self.c.halo_pack(side, self.sbuf[side].data_ptr(), self._s(self.compute))
        return self.sbuf[side]

    def recv_buffer(self, side):
        return self.rbuf[side]

    def comm_begin(self):
        # the collective library orders its work after the *current* stream: make that
        # the comm stream, which waits for the pack kernels, so the interior step that
        # is launched next on the compute stream runs concurrently with the exchange
        self.comm.wait_stream(self.compute)
        self._ctx = self.torch.cuda.stream(self.comm)
        self._ctx.__enter__()

    def comm_end(self):
        if self._ctx is not None:
            self._ctx.__exit__(None, None, None)
            self._ctx = None
        self.compute.wait_stream(self.comm)

    def unpack(self, side):
        self.c.halo_unpack(side, self.rbuf[side].data_ptr(), self._s(self.compute))

    def step_interior(self, dt):
        if self._ctx is not None:           # leave the comm-stream context for compute work
            self._ctx.__exit__(None, None, None)
            self._ctx = None
        self.c.step_interior(dt, self._s(self.compute))

    def step_boundary(self, dt):
        self.c.step_boundary(dt, self._s(self.compute))

    def step_all(self, dt):
        self.c.step(1, dt)
