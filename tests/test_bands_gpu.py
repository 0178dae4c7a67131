"""GPU tests of the latitude-band path: (A) several bands of one grid stepped in ONE process
on one GPU with the ghost rows moved by device copies -- exercises the ghost-row kernels,
gcm_halo_pack/unpack and gcm_step_interior/boundary against the single-band result; (B) two
processes sharing the GPU, BandRunner + HipBandEngine over torch.distributed (gloo here: RCCL
refuses two ranks on one device; the 8-GPU run uses the same code with backend nccl)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _ic2d(shape):
    rng = np.random.default_rng(11)
    return dict(u=rng.standard_normal(shape), v=rng.standard_normal(shape),
                p=101325 + rng.standard_normal(shape), t=273.16 + rng.standard_normal(shape),
                q=rng.random(shape))


def _ic_pe(geom):
    rng = np.random.default_rng(12)
    L, H, W = geom.layers, geom.height, geom.width
    p = 1e5 + 10 * rng.standard_normal((H, W))
    u = rng.standard_normal((L, H, W))
    v = rng.standard_normal((L, H, W))
    v[:, -1, :] = 0
    sig = np.asarray(geom.sig)
    t = (300 + rng.standard_normal((L, H, W))) * ((1e5 / (p * sig + geom.ptop)) ** (287.0 / 1004.0))
    q = 3e-6 * (1 + 0.1 * rng.random((L, H, W)))
    return p, u, v, t, q


UTC0 = 5 * 3600.0


def _ic_gt(H, W):
    return 288.0 + np.random.default_rng(13).standard_normal((H, W))


def _exchange(cores, torch):
    """ring exchange by device copies: the rows a band packs on side s land in the neighbour's
    opposite ghost"""
    n = len(cores)
    bufs = [[torch.empty(c.halo_bytes() // 8, dtype=torch.float64, device="cuda") for _ in (0, 1)]
            for c in cores]
    for r, c in enumerate(cores):
        c.halo_pack(0, bufs[r][0].data_ptr())
        c.halo_pack(1, bufs[r][1].data_ptr())
    torch.cuda.synchronize()
    for r, c in enumerate(cores):
        c.halo_unpack(1, bufs[(r + 1) % n][0].data_ptr())   # south ghost <- southern band's north edge
        c.halo_unpack(0, bufs[(r - 1) % n][1].data_ptr())   # north ghost <- northern band's south edge
    torch.cuda.synchronize()


@pytest.mark.parametrize("variant", ["fused", "staged"])
@pytest.mark.parametrize("nb", [2, 3])
def test_bands_in_process_2d(variant, nb):
    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd.bands import split_rows
    H, W, steps = 37, 130, 3
    var = g._lib.VARIANT_FUSED if variant == "fused" else g._lib.VARIANT_STAGED
    f = _ic2d((H, W))
    ref = g.Core(g._lib.SW2D_TEMP, W, H, dx=300e3, tracer=g._lib.TRACER_VANLEER, variant=var)
    ref.set_state(**f)
    ref.step(steps, 300.0)
    want = ref.get_state()
    ref.close()
    cores = []
    for r, (row0, n) in enumerate(split_rows(H, nb)):
        c = g.Core(g._lib.SW2D_TEMP, W, n, dx=300e3, tracer=g._lib.TRACER_VANLEER, variant=var,
                   nranks=nb, rank=r, global_height=H, row0=row0)
        c.set_state(**{k: v[row0:row0 + n] for k, v in f.items()})
        cores.append(c)
    for _ in range(steps):
        _exchange(cores, torch)
        for c in cores:
            c.step_interior(300.0)
        for c in cores:
            c.step_boundary(300.0)
    got = [np.concatenate(x, axis=0) for x in zip(*[c.get_state() for c in cores])]
    for c in cores:
        c.close()
    for k, a, b in zip("puvtq", got, want):
        assert rel_err(a, b) < 1e-13, (k, rel_err(a, b))


@pytest.mark.parametrize("variant", ["fused", "staged"])
def test_deep_halo_in_process_2d(variant):
    """halo_steps = 3: 6 ghost rows per side, one exchange per 3 steps"""
    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd.bands import split_rows
    H, W, k, nb = 40, 130, 3, 3
    var = g._lib.VARIANT_FUSED if variant == "fused" else g._lib.VARIANT_STAGED
    f = _ic2d((H, W))
    ref = g.Core(g._lib.SW2D_TEMP, W, H, dx=300e3, tracer=g._lib.TRACER_VANLEER, variant=var)
    ref.set_state(**f)
    ref.step(2 * k, 300.0)
    want = ref.get_state()
    ref.close()
    cores = []
    for r, (row0, n) in enumerate(split_rows(H, nb)):
        c = g.Core(g._lib.SW2D_TEMP, W, n, dx=300e3, tracer=g._lib.TRACER_VANLEER, variant=var,
                   nranks=nb, rank=r, global_height=H, row0=row0, halo_steps=k)
        c.set_state(**{kk: v[row0:row0 + n] for kk, v in f.items()})
        cores.append(c)
    for _ in range(2):
        _exchange(cores, torch)
        for c in cores:
            c.step(k, 300.0)
        with pytest.raises(g.GcmError):
            cores[0].step(1, 300.0)          # ghost rows used up: an exchange is due
    got = [np.concatenate(x, axis=0) for x in zip(*[c.get_state() for c in cores])]
    for c in cores:
        c.close()
    for kk, a, b in zip("puvtq", got, want):
        assert rel_err(a, b) < 1e-13, (kk, rel_err(a, b))


@pytest.mark.parametrize("nb", [2, 3])
def test_bands_in_process_pe25d(nb):
    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    from gcmiipy_amd.bands import split_rows
    H, W, L, steps = 14, 20, 5, 2
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    geom.heightmap[H // 2, 3] = 300.0
    ic = _ic_pe(geom)
    ref = g.Core(g._lib.PE25D, W, H, L, geom=geom)
    ref.set_state(*ic)
    ref.step(steps, 120.0)
    want = ref.get_state()
    ref.close()
    cores = []
    for r, (row0, n) in enumerate(split_rows(H, nb)):
        c = g.Core(g._lib.PE25D, W, n, L, geom=geom, nranks=nb, rank=r, global_height=H, row0=row0)
        sl = slice(row0, row0 + n)
        c.set_state(ic[0][sl], *[a[:, sl] for a in ic[1:]])
        cores.append(c)
    for _ in range(steps):
        _exchange(cores, torch)          # current state
        for c in cores:
            c.step_interior(120.0)       # predictor
        _exchange(cores, torch)          # predicted state
        for c in cores:
            c.step_boundary(120.0)       # corrector
    parts = [c.get_state() for c in cores]
    for c in cores:
        c.close()
    for f, k in enumerate("puvtq"):
        got = np.concatenate([p_[f] for p_ in parts], axis=0 if f == 0 else 1)
        assert rel_err(got, want[f]) < 1e-13, (k, rel_err(got, want[f]))


@pytest.mark.parametrize("nb", [2, 3])
def test_bands_in_process_pe25d_edge_first(nb):
    """gcm_step_phase: edge rows first, ghost exchange, interior rows -- same numbers as one band"""
    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    from gcmiipy_amd.bands import split_rows
    H, W, L, steps = 15, 20, 5, 3
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    ic = _ic_pe(geom)
    ref = g.Core(g._lib.PE25D, W, H, L, geom=geom)
    ref.set_state(*ic)
    ref.step(steps, 120.0)
    want = ref.get_state()
    ref.close()
    cores = []
    for r, (row0, n) in enumerate(split_rows(H, nb)):
        c = g.Core(g._lib.PE25D, W, n, L, geom=geom, nranks=nb, rank=r, global_height=H, row0=row0)
        sl = slice(row0, row0 + n)
        c.set_state(ic[0][sl], *[a[:, sl] for a in ic[1:]])
        cores.append(c)
    _exchange(cores, torch)                  # ghosts of the initial state
    for _ in range(steps):
        for stage in (0, 1):
            for c in cores:
                c.step_phase(2 * stage, 120.0)
            bufs = [[torch.empty(c.halo_bytes() // 8, dtype=torch.float64, device="cuda") for _ in (0, 1)]
                    for c in cores]
            for r, c in enumerate(cores):        # pack BEFORE the interior phase, as the runner does
                c.halo_pack(0, bufs[r][0].data_ptr())
                c.halo_pack(1, bufs[r][1].data_ptr())
            for c in cores:
                c.step_phase(2 * stage + 1, 120.0)
            torch.cuda.synchronize()
            for r, c in enumerate(cores):
                c.halo_unpack(1, bufs[(r + 1) % nb][0].data_ptr())
                c.halo_unpack(0, bufs[(r - 1) % nb][1].data_ptr())
            torch.cuda.synchronize()
    parts = [c.get_state() for c in cores]
    for c in cores:
        c.close()
    for f, k in enumerate("puvtq"):
        got = np.concatenate([p_[f] for p_ in parts], axis=0 if f == 0 else 1)
        assert rel_err(got, want[f]) < 1e-13, (k, rel_err(got, want[f]))


class _Ring:
    """torch.distributed stand-in for several band engines in ONE process: a posted exchange is
    carried out, stream-ordered and without any host synchronisation, when its requests are waited
    for -- by then every rank has posted.  A receive waits (GPU side) for the sender's post event,
    i.e. for the sender's comm stream at the point where its send buffers are valid."""
    isend, irecv = "isend", "irecv"

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    def __init__(self, torch, hub, rank):
        self.torch, self.hub, self.rank = torch, hub, rank

    def batch_isend_irecv(self, ops):
        stream = self.torch.cuda.current_stream()
        ev = self.torch.cuda.Event()
        ev.record(stream)
        self.hub[self.rank] = (ops, ev, stream)
        ring = self

        class Req:
            def wait(self):
                ops_, _, stream_ = ring.hub[ring.rank]
                recvs = [o for o in ops_ if o.op == "irecv"]
                with ring.torch.cuda.stream(stream_):
                    for n, o in enumerate(recvs):          # n-th receive <- the peer's n-th send
                        pops, pev, _ = ring.hub[o.peer]
                        sends = [q for q in pops if q.op == "isend" and q.peer == ring.rank]
                        src = sends[n] if len(sends) > 1 else sends[0]
                        stream_.wait_event(pev)
                        o.tensor.copy_(src.tensor, non_blocking=True)
                    done = ring.torch.cuda.Event()
                    done.record(stream_)
                    ring.hub.setdefault("done", []).append(done)   # = the peers' sends completed
        return [Req()]


@pytest.mark.parametrize("nb", [2, 3, 4])
def test_band_runners_in_process_stream_ordered(nb):
    """BandRunner + HipBandEngine for GCM_PE25D, every band on its own compute and comm stream,
    exchange = stream-ordered device copies: no host synchronisation anywhere inside the steps,
    so a missing stream dependency (edge rows updated and packed on the library's second stream
    while the interior rows run) shows up as wrong numbers.  Bit-identical to the single band."""
    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    from gcmiipy_amd.bands import BandRunner, HipBandEngine, split_rows
    H, W, L, steps = 23, 36, 9, 4
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    ic = _ic_pe(geom)
    ref = g.Core(g._lib.PE25D, W, H, L, geom=geom)
    ref.set_state(*ic)
    ref.step(steps, 120.0)
    want = ref.get_state()
    ref.close()
    hub, runners, cores = {}, [], []
    for r, (row0, n) in enumerate(split_rows(H, nb)):
        c = g.Core(g._lib.PE25D, W, n, L, geom=geom, nranks=nb, rank=r, global_height=H, row0=row0)
        sl = slice(row0, row0 + n)
        c.set_state(ic[0][sl], *[a[:, sl] for a in ic[1:]])
        with torch.cuda.stream(torch.cuda.Stream()):
            eng = HipBandEngine(c, torch)
        assert eng.edge_first and eng.async_edges
        runners.append(BandRunner(eng, r, nb, _Ring(torch, hub, r)))
        cores.append(c)

    def all_ranks(post):
        posted = []
        for rn in runners:
            posted.append(post(rn))
            rn.e._leave_comm()              # several engines in one thread: do not nest stream contexts
        for rn, reqs in zip(runners, posted):
            rn._finish(reqs)
        for rn in runners:                  # a send buffer is reused only after its send completed
            for ev in hub.get("done", []):
                rn.e.compute.wait_event(ev)
        hub["done"] = []

    all_ranks(lambda rn: rn.exchange_start())            # ghosts of the initial state
    for _ in range(steps):
        for stage in (0, 1):
            all_ranks(lambda rn: rn.begin_stage(stage, 120.0))
    torch.cuda.synchronize()
    parts = [c.get_state() for c in cores]
    for c in cores:
        c.close()
    for f, k in enumerate("puvtq"):
        got = np.concatenate([p_[f] for p_ in parts], axis=0 if f == 0 else 1)
        assert np.array_equal(got, want[f]), (k, rel_err(got, want[f]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, model, outdir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    from gcmiipy_amd.bands import BandRunner, HipBandEngine, split_rows
    torch.cuda.set_device(0)
    # rendezvous through a file in the test's own directory: no TCP port to collide on
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank, world_size=world)
    if model in ("c3", "c3deep"):
        H, W, steps, dt = 40, 130, 3 if model == "c3" else 4, 300.0
        f = _ic2d((H, W))
        row0, n = split_rows(H, world)[rank]
        c = g.Core(g._lib.SW2D_TEMP, W, n, dx=300e3, tracer=g._lib.TRACER_VANLEER, nranks=world, rank=rank,
                   global_height=H, row0=row0, stream=torch.cuda.current_stream().cuda_stream,
                   halo_steps=1 if model == "c3" else 2)
        c.set_state(**{k: v[row0:row0 + n] for k, v in f.items()})
    else:
        H, W, L, steps, dt = 14, 20, 5, 2 if model == "pe" else 3, 120.0
        geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
        ic = _ic_pe(geom)
        row0, n = split_rows(H, world)[rank]
        sl = slice(row0, row0 + n)
        c = g.Core(g._lib.PE25D, W, n, L, geom=geom, nranks=world, rank=rank, global_height=H, row0=row0,
                   stream=torch.cuda.current_stream().cuda_stream)
        c.set_state(ic[0][sl], *[a[:, sl] for a in ic[1:]])
    eng = HipBandEngine(c, torch, stream_aware=False)
    if model == "pephys":             # BASELINE configs[4]'s second phase on bands that are separate processes
        c.set_ground(_ic_gt(H, W)[sl])
        eng.set_physics(geom, UTC0)
    runner = BandRunner(eng, rank, world, dist)
    runner.run(1, dt)                 # chunked path: a partial window first, then the rest
    runner.run(steps - 1, dt)
    torch.cuda.synchronize()
    st = c.get_state() + ([c.get_ground()] if model == "pephys" else [])
    np.savez(os.path.join(outdir, "r%d.npz" % rank), **{k: a for k, a in zip("puvtqg", st) if a is not None})
    c.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("model,world", [("c3", 2), ("c3deep", 2), ("pe", 2), ("c3deep", 4), ("pe", 4), ("pephys", 2), ("pephys", 3)])
def test_band_runner_processes_one_gpu(tmp_path, model, world):
    """world = 4: every rank has two distinct ring neighbours (the N = 2 ring talks to one peer
    twice); four processes share the GPU (the box allows six)"""
    import torch.multiprocessing as mp
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    mp.spawn(_worker, args=(world, _free_port(), model, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    if model in ("c3", "c3deep"):
        H, W = 40, 130
        ref = g.Core(g._lib.SW2D_TEMP, W, H, dx=300e3, tracer=g._lib.TRACER_VANLEER)
        ref.set_state(**_ic2d((H, W)))
        ref.step(3 if model == "c3" else 4, 300.0)
    else:
        H, W, L = 14, 20, 5
        geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
        ref = g.Core(g._lib.PE25D, W, H, L, geom=geom)
        ref.set_state(*_ic_pe(geom))
        if model == "pephys":
            ref.set_ground(_ic_gt(H, W))
            ref.set_physics(geom, UTC0)
        ref.step(2 if model == "pe" else 3, 120.0)
    want = ref.get_state() + ([ref.get_ground()] if model == "pephys" else [])
    ref.close()
    for f, k in enumerate("puvtqg"[:len(want)]):
        axis = 1 if (model in ("pe", "pephys") and 0 < f < 5) else 0
        got = np.concatenate([p_[k] for p_ in parts], axis=axis)
        assert rel_err(got, want[f]) < 1e-13, (k, rel_err(got, want[f]))


def test_halo_buffer_registration_errors():
    """gcm_set_halo_buffers / gcm_wait_edges argument and state checking"""
    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    from gcmiipy_amd.core import GcmError
    H, W, L = 12, 16, 4
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    single = g.Core(g._lib.PE25D, W, H, L, geom=geom)
    buf = torch.empty(1 << 16, dtype=torch.float64, device="cuda")
    with pytest.raises(GcmError, match="not a latitude band"):
        single.set_halo_buffers(buf.data_ptr(), buf.data_ptr())
    single.close()
    band = g.Core(g._lib.PE25D, W, 6, L, geom=geom, nranks=2, rank=0, global_height=H, row0=0)
    with pytest.raises(GcmError, match="no send buffers registered"):
        band.wait_edges(None)
    with pytest.raises(ValueError, match="both buffers"):
        band.set_halo_buffers(buf.data_ptr(), None)
    band.set_halo_buffers(buf.data_ptr(), buf[1 << 15:].data_ptr())
    band.wait_edges(None)                      # an event that was never recorded: returns at once
    band.set_halo_buffers(None, None)          # unregister
    with pytest.raises(GcmError, match="no send buffers registered"):
        band.wait_edges(None)
    band.close()
    sw = g.Core(g._lib.SW2D, 32, 8, dx=300e3, nranks=2, rank=0, global_height=16, row0=0)
    with pytest.raises(GcmError, match="GCM_PE25D latitude bands only"):
        sw.set_halo_buffers(buf.data_ptr(), buf.data_ptr())
    sw.close()


def _rccl_self_worker(rank, port, model, outdir):
    """one process, one GPU, the REAL backend: an RCCL ("nccl") group of one rank whose band is its
    own north and south neighbour -- numerically the periodic single domain"""
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    from gcmiipy_amd.bands import BandRunner, HipBandEngine
    torch.cuda.set_device(0)
    direct = "-direct" in model                 # gcmiipy_amd.rccl: RCCL called through ctypes
    if model.endswith("-hostloop"):             # the host-driven sequence instead of gcm_band_run
        os.environ["GCM_BAND_HOST_LOOP"] = "1"
    model = model.split("-")[0]
    dist.init_process_group("nccl", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    if direct:
        from gcmiipy_amd.rccl import RcclP2P
        ring = RcclP2P(None, 0, 1, uid_bytes=RcclP2P.new_unique_id())   # bench.py's bring-up path
        ring.self_check()
        assert ring.count() == 1                   # ncclCommCount: what bench.py prints as rccl_ranks
    else:
        ring = dist
    phys = model == "pephys"
    if model in ("pe", "pephys"):
        H, W, L, steps, dt = 23, 36, 9, 5, 120.0
        geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
        c = g.Core(g._lib.PE25D, W, H, L, geom=geom, nranks=2, rank=0, global_height=H, row0=0,
                   stream=torch.cuda.current_stream().cuda_stream)
        c.set_state(*_ic_pe(geom))
        if phys:
            c.set_ground(_ic_gt(H, W))
    else:
        H, W, steps, dt = 64, 130, 11, 300.0
        c = g.Core(g._lib.SW2D_TEMP, W, H, dx=300e3, tracer=g._lib.TRACER_VANLEER, nranks=2, rank=0,
                   global_height=H, row0=0, stream=torch.cuda.current_stream().cuda_stream,
                   halo_steps=1 if model == "c3" else 4)
        c.set_state(**_ic2d((H, W)))
    eng = HipBandEngine(c, torch)                 # stream-aware: the exchange is ordered on streams only
    if phys:
        eng.set_physics(geom, UTC0)
    runner = BandRunner(eng, 0, 2, ring, north=0, south=0)
    assert runner.native == (direct and "GCM_BAND_HOST_LOOP" not in os.environ)
    runner.run(steps - 2, dt)
    runner.run(2, dt)                             # a second call continues on the exchanged ghosts
    torch.cuda.synchronize()
    st = c.get_state() + ([c.get_ground()] if phys else [])
    np.savez(os.path.join(outdir, "self.npz"), **{k: a for k, a in zip("puvtqg", st) if a is not None})
    c.close()
    if direct:
        ring.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("model", ["pe-direct", "c3-direct", "c3deep-direct", "pe-direct-hostloop",
                                   "c3deep-direct-hostloop", "pe", "c3deep", "pephys-direct", "pephys-direct-hostloop"])
def test_band_runner_over_rccl_self_ring(tmp_path, model):
    """The production exchange paths on the one GPU of the test box -- RCCL called directly
    (gcmiipy_amd.rccl: ncclSend/ncclRecv groups on the library's comm stream, posted by the LIBRARY
    inside gcm_band_run, one call per run: what bench.py uses; "-hostloop": the same exchange posted
    from Python step by step)
    and through torch.distributed "nccl" (batch_isend_irecv) -- with no host synchronisation.  RCCL
    refuses two ranks per device, but a rank may send to itself, and a band that is its own
    neighbour on both sides is the periodic single domain.  Bit-identical to it (GCM_PE25D: edge
    rows updated and packed on the library's second stream while the interior rows run)."""
    import torch.multiprocessing as mp
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    mp.spawn(_rccl_self_worker, args=(_free_port(), model, str(tmp_path)), nprocs=1, join=True)
    got = np.load(os.path.join(str(tmp_path), "self.npz"))
    model = model.split("-")[0]
    if model in ("pe", "pephys"):
        H, W, L = 23, 36, 9
        geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
        ref = g.Core(g._lib.PE25D, W, H, L, geom=geom)
        ref.set_state(*_ic_pe(geom))
        if model == "pephys":                      # the single domain: gcm_step with the physics registered
            ref.set_ground(_ic_gt(H, W))
            ref.set_physics(geom, UTC0)
        ref.step(5, 120.0)
    else:
        H, W = 64, 130
        ref = g.Core(g._lib.SW2D_TEMP, W, H, dx=300e3, tracer=g._lib.TRACER_VANLEER)
        ref.set_state(**_ic2d((H, W)))
        ref.step(11, 300.0)
    want = ref.get_state() + ([ref.get_ground()] if model == "pephys" else [])
    ref.close()
    for k, w in zip("puvtqg", want):
        assert np.array_equal(got[k], w), k


@pytest.mark.parametrize("model", ["pe", "c3", "c3deep", "c3deep-overlap", "c3deep8-overlap", "pephys"])
def test_band_run_native_loopback_equals_single_domain(model):
    """gcm_band_run with the loopback exchange (gcm_set_exchange without RCCL entry points): the band
    is its own neighbour, i.e. the periodic single domain -- bit-identical to it, and to the
    host-driven sequence.  "-overlap": the deep-halo exchange hidden behind the interior rows of the
    steps around it (gcm_set_band_overlap), runs cut at every position of a window.  Also the error
    paths of gcm_set_exchange / gcm_band_run."""
    overlap = model.endswith("-overlap")
    halo = 8 if model.startswith("c3deep8") else 4 if model.startswith("c3deep") else 1
    model = model.split("-")[0].replace("8", "")
    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    from gcmiipy_amd.bands import BandRunner, HipBandEngine, LoopbackExchange
    from gcmiipy_amd.core import GcmError
    phys = model == "pephys"
    if model in ("pe", "pephys"):
        H, W, L, steps, dt = 23, 36, 9, 5, 120.0
        geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
        mk = lambda **kw: g.Core(g._lib.PE25D, W, H, L, geom=geom, **kw)
        ic = dict(zip(("p", "u", "v", "t", "q"), _ic_pe(geom)))
    else:
        H, W, steps, dt = 64, 130, 11 if not overlap else 23, 300.0
        mk = lambda **kw: g.Core(g._lib.SW2D_TEMP, W, H, dx=300e3, tracer=g._lib.TRACER_VANLEER, **kw)
        ic = _ic2d((H, W))
    ref = mk()
    ref.set_state(**ic)
    if phys:
        # explicit calls on the single domain: step, solar_step at the clock, clock += dt (no_limits_2_5d.py:229-234)
        ref.set_ground(_ic_gt(H, W))
        for n in range(steps):
            ref.step(1, dt)
            ref.solar_step(geom, dt, UTC0 + n * dt)
    else:
        ref.step(steps, dt)
    want = ref.get_state() + ([ref.get_ground()] if phys else [])
    with pytest.raises(GcmError, match="not a latitude band"):
        ref.band_run(1, dt)
    ref.close()
    c = mk(nranks=2, rank=0, global_height=H, row0=0, stream=torch.cuda.current_stream().cuda_stream,
           halo_steps=halo)
    c.set_state(**ic)
    with pytest.raises(GcmError, match="no exchange registered"):
        c.band_run(1, dt)
    eng = HipBandEngine(c, torch)
    if phys:
        c.set_ground(_ic_gt(H, W))
        eng.set_physics(geom, UTC0)
    runner = BandRunner(eng, 0, 2, LoopbackExchange(), north=0, south=0)
    assert runner.native
    if overlap:
        c.set_band_overlap(True)
        for n in (1, 2, 3, 5, 4, 8):                    # ends inside, at the end and at the start of a window
            runner.run(n, dt)
    else:
        runner.run(3, dt)
        runner.run(steps - 3, dt)
    torch.cuda.synchronize()
    got = c.get_state() + ([c.get_ground()] if phys else [])
    if phys:
        assert c.utc() == UTC0 + steps * dt
    c.close()
    for k, a, b in zip("puvtqg", got, want):
        assert (a is None and b is None) or np.array_equal(a, b), k


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_eight_bands_with_physics_equal_single_domain(dtype):
    """BASELINE configs[4] on its stated decomposition: 8 latitude bands of a (40, 64, 2880) grid, dynamics +
    solar_timestep for 3 steps, the bands stepped in ONE process with the ghost rows moved by device copies.  The
    radiation changes theta and the ground temperature after the post-corrector exchange; every band radiates its
    ghost rows locally (the ghost rows of gt travel in the messages), so the next predictor reads current values.
    Bit for bit the single domain, ground temperature included."""
    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    from gcmiipy_amd.bands import split_rows
    H, W, L, steps, nb, dt = 64, 2880, 40, 3, 8, 60.0
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    ic, gt = _ic_pe(geom), _ic_gt(H, W)
    ref = g.Core(g._lib.PE25D, W, H, L, geom=geom, dtype=dtype)
    ref.set_state(*ic)
    ref.set_ground(gt)
    dyn_only = None
    for n in range(steps):
        ref.step(1, dt)
        if n == 0:
            dyn_only = ref.get_state((3,))[3]
        ref.solar_step(geom, dt, UTC0 + n * dt)
    want, want_gt = ref.get_state(), ref.get_ground()
    ref.close()
    cores = []
    for r, (row0, n) in enumerate(split_rows(H, nb)):
        c = g.Core(g._lib.PE25D, W, n, L, geom=geom, nranks=nb, rank=r, global_height=H, row0=row0, dtype=dtype)
        sl = slice(row0, row0 + n)
        c.set_state(ic[0][sl], *[a[:, sl] for a in ic[1:]])
        c.set_ground(gt[sl])
        cores.append(c)
    bufs = [[torch.empty(c.halo_bytes(), dtype=torch.uint8, device="cuda") for _ in (0, 1)] for c in cores]

    def exchange():
        for r, c in enumerate(cores):
            c.halo_pack(0, bufs[r][0].data_ptr())
            c.halo_pack(1, bufs[r][1].data_ptr())
        torch.cuda.synchronize()
        for r, c in enumerate(cores):
            c.halo_unpack(1, bufs[(r + 1) % nb][0].data_ptr())
            c.halo_unpack(0, bufs[(r - 1) % nb][1].data_ptr())
        torch.cuda.synchronize()
    exchange()                               # once: the ghost rows of the initial state and of gt
    for n in range(steps):                   # the order of gcm_band_run: two exchanges per step, none after the physics
        for c in cores:
            c.step_interior(dt)              # predictor
        exchange()                           # predicted state
        for c in cores:
            c.step_boundary(dt)              # corrector
        exchange()                           # new state, BEFORE the radiation changes theta
        for c in cores:
            c.solar_step(geom, dt, UTC0 + n * dt)      # own rows and ghost rows
    parts = [c.get_state() for c in cores]
    got_gt = np.concatenate([c.get_ground() for c in cores], axis=0)
    for c in cores:
        c.close()
    for f in range(5):
        got = np.concatenate([x[f] for x in parts], axis=0 if f == 0 else 1)
        assert np.array_equal(got, want[f]), "puvtq"[f]
    assert np.array_equal(got_gt, want_gt)
    assert not np.array_equal(dyn_only, want[3])


@pytest.mark.parametrize("phys", [False, True])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_band_run_chains_at_overlapping_size(dtype, phys, monkeypatch):
    """gcm_band_run keeps two chains of launches on two streams with no join per stage; at the small sizes of the
    other band tests a kernel is over before the other chain starts, so a missing dependency could not show.  Here
    the band is 48 rows x 1440 columns x 24 levels (edge rows marched in level segments, kernels of tens of
    microseconds on either stream), five steps inside gcm_band_run calls, fp64 and fp32, with the loopback exchange
    (the band is its own neighbour = the periodic single domain).  Four orchestrations -- the chains of the product, one
    stream only (GCM_PE_SINGLE_STREAM=1), the exchange on the comm stream with a join per stage
    (GCM_BAND_COMM_STREAM=1), the edge rows' update dispatched ahead of the interior rows' (GCM_BAND_OVERLAP=1) --
    must all give the single domain's bits; a solar_timestep and a gcm_set_state between
    runs exercise the reset of the queued ghost-row work (ghost_ready)."""
    import torch
    import gcmiipy_amd as g
    from gcmiipy_amd import geometry
    from gcmiipy_amd.bands import BandRunner, HipBandEngine, LoopbackExchange
    H, W, L, dt = 48, 1440, 24, 1.0
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    ic, gt = _ic_pe(geom), _ic_gt(H, W)

    def drive(core, run):
        """2 steps, an explicit solar_timestep, 3 steps; then the initial state again and 2 steps"""
        core.set_state(*ic)
        core.set_ground(gt)
        run(2)
        core.set_ground(core.get_ground())        # the same values again: a band exchanges its ghost rows anew (on the caller's stream)
        core.solar_step(geom, dt, 7 * 3600.0)
        run(3)
        a = core.get_state() + [core.get_ground()]
        core.set_state(*ic)
        core.set_ground(gt)
        if phys:
            core.set_physics(geom, UTC0)          # the clock starts again with the state
        run(2)
        return a, core.get_state() + [core.get_ground()]

    ref = g.Core(g._lib.PE25D, W, H, L, geom=geom, dtype=dtype)
    if phys:
        ref.set_physics(geom, UTC0)
    want = drive(ref, lambda n: ref.step(n, dt))
    ref.close()
    # (GCM_BAND_OVERLAP=1 = gcm_set_band_overlap(1): the interior rows' update held back until the edge rows' is dispatched)
    for env in ({}, {"GCM_PE_SINGLE_STREAM": "1"}, {"GCM_BAND_COMM_STREAM": "1"}, {"GCM_BAND_OVERLAP": "1"}):
        for k in ("GCM_PE_SINGLE_STREAM", "GCM_BAND_COMM_STREAM", "GCM_BAND_OVERLAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = g.Core(g._lib.PE25D, W, H, L, geom=geom, nranks=2, rank=0, global_height=H, row0=0, dtype=dtype,
                   stream=torch.cuda.current_stream().cuda_stream)
        eng = HipBandEngine(c, torch)
        if phys:
            eng.set_physics(geom, UTC0)
        runner = BandRunner(eng, 0, 2, LoopbackExchange(), north=0, south=0)
        assert runner.native

        def run(n):
            runner.run(n, dt)
            torch.cuda.synchronize()
        got = drive(c, run)
        c.close()
        for part, (ga, wa) in enumerate(zip(got, want)):
            for k, a, b in zip("puvtqg", ga, wa):
                assert np.array_equal(a, b), (env, part, k, rel_err(a, b))
