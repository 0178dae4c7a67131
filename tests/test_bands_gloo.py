"""N > 1 path on the CPU: gcmiipy_amd.bands.BandRunner over torch.distributed/gloo with
world_size 2 and 3, NumPy band engines (tests/band_engines.py).  The banded result must
equal the single-domain oracle BIT FOR BIT (same arithmetic per row)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _ic2d(shape, temp):
    rng = np.random.default_rng(5)
    f = dict(u=rng.standard_normal(shape), v=rng.standard_normal(shape))
    if temp:
        f.update(p=101325 + rng.standard_normal(shape), t=273.16 + rng.standard_normal(shape),
                 q=rng.random(shape))
    else:
        f.update(p=8000 + rng.standard_normal(shape))
    return f


def _worker_2d(rank, world, port, shape, temp, steps, outdir, halo_steps=1):
    from band_engines import NumpyBand2D
    from gcmiipy_amd.bands import BandRunner, split_rows
    # rendezvous through a file in the test's own directory: no TCP port to collide on
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank, world_size=world)
    full = _ic2d(shape, temp)
    row0, n = split_rows(shape[0], world)[rank]
    log = []
    eng = NumpyBand2D({k: v[row0:row0 + n] for k, v in full.items()}, 300e3, temp, log, halo_steps)
    runner = BandRunner(eng, rank, world, dist)
    for _ in range(steps):
        runner.step(300.0)
    if halo_steps == 1:
        # the runner must start the exchange, then the interior, then wait/unpack, then the boundary
        assert log[:6] == ["comm_begin", "interior", "comm_end", "unpack0", "unpack1", "boundary"], log[:6]
    else:
        assert log.count("unpack0") == -(-steps // halo_steps)      # one exchange per halo_steps steps
    np.savez(os.path.join(outdir, "r%d.npz" % rank), **eng.interior_state())
    dist.barrier()
    dist.destroy_process_group()


def _worker_pe(rank, world, port, hwl, steps, outdir, edge_first=False, phys=False):
    from band_engines import NumpyBandPE
    from gcmiipy_amd.bands import BandRunner, split_rows
    from oracle import geometry as ogeo
    # rendezvous through a file in the test's own directory: no TCP port to collide on
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "rendezvous"), rank=rank, world_size=world)
    H, W, L = hwl
    geom = ogeo.gen_geometry(H, W, L, sig_func=ogeo.manabe_sig)
    geom.heightmap[H // 2, 3] = 300.0
    p, u, v, t, q = _ic_pe(geom)
    row0, n = split_rows(H, world)[rank]
    sl = slice(row0, row0 + n)
    gt = _ic_gt(geom)[sl] if phys else None
    eng = NumpyBandPE(p[sl], u[:, sl], v[:, sl], t[:, sl], q[:, sl], geom, row0, edge_first, gt=gt, utc=UTC0)
    runner = BandRunner(eng, rank, world, dist)
    for _ in range(steps):
        runner.step(120.0)
    st = eng.interior_state()
    np.savez(os.path.join(outdir, "r%d.npz" % rank), **dict(zip("puvtqg", st)))
    dist.barrier()
    dist.destroy_process_group()


def _ic_pe(geom):
    from oracle import temperature
    rng = np.random.default_rng(9)
    L, H, W = geom.layers, geom.height, geom.width
    p = 1e5 + 10 * rng.standard_normal((H, W))
    u = rng.standard_normal((L, H, W))
    v = rng.standard_normal((L, H, W))
    v[:, -1, :] = 0
    t = temperature.to_potential_temp(300 + rng.standard_normal((L, H, W)), p * geom.sig + geom.ptop)
    q = 3e-6 * (1 + 0.1 * rng.random((L, H, W)))
    return p, u, v, t, q


UTC0 = 5 * 3600.0


def _ic_gt(geom):
    return 288.0 + np.random.default_rng(10).standard_normal((geom.height, geom.width))


def test_split_rows():
    from gcmiipy_amd.bands import split_rows
    assert split_rows(720, 8) == [(90 * r, 90) for r in range(8)]
    assert split_rows(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert sum(n for _, n in split_rows(2048, 7)) == 2048


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("temp", [False, True])
def test_banded_2d_equals_single_domain(tmp_path, world, temp):
    from oracle import sw2d, sw2d_temp, tracer
    shape, steps = (14, 24), 3
    mp.spawn(_worker_2d, args=(world, _free_port(), shape, temp, steps, str(tmp_path)), nprocs=world,
             join=True)
    f = _ic2d(shape, temp)
    for _ in range(steps):
        if temp:
            q = tracer.limited_advection(300.0, (300e3, 300e3), np.stack([f["v"], f["u"]]), f["q"])
            u, v, p, t = sw2d_temp.matsumo_scheme(f["u"], f["v"], f["p"], f["t"], 300e3, 300.0)
            f = dict(u=u, v=v, p=p, t=t, q=q)
        else:
            u, v, p = sw2d.matsumo_scheme(f["u"], f["v"], f["p"], 300e3, 300.0)
            f = dict(u=u, v=v, p=p)
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    for k in f:
        got = np.concatenate([pp[k] for pp in parts], axis=0)
        assert np.array_equal(got, f[k]), k


def test_banded_2d_deep_halo(tmp_path):
    """halo_steps = 2: exchange every second step, 5 steps (last window partial)"""
    from oracle import sw2d
    shape, steps, world = (18, 24), 5, 3
    mp.spawn(_worker_2d, args=(world, _free_port(), shape, False, steps, str(tmp_path), 2), nprocs=world,
             join=True)
    f = _ic2d(shape, False)
    for _ in range(steps):
        u, v, p = sw2d.matsumo_scheme(f["u"], f["v"], f["p"], 300e3, 300.0)
        f = dict(u=u, v=v, p=p)
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    for k in f:
        assert np.array_equal(np.concatenate([pp[k] for pp in parts], axis=0), f[k]), k


@pytest.mark.parametrize("edge_first", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_banded_pe25d_equals_single_domain(tmp_path, world, edge_first):
    from oracle import dynamics, geometry as ogeo
    hwl, steps = (12, 16, 3), 2
    mp.spawn(_worker_pe, args=(world, _free_port(), hwl, steps, str(tmp_path), edge_first), nprocs=world,
             join=True)
    H, W, L = hwl
    geom = ogeo.gen_geometry(H, W, L, sig_func=ogeo.manabe_sig)
    geom.heightmap[H // 2, 3] = 300.0
    st = _ic_pe(geom)
    for _ in range(steps):
        st = dynamics.matsuno_timestep(*st, 120.0, geom)
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    for k, want in zip("puvtq", st):
        got = np.concatenate([pp[k] for pp in parts], axis=0 if k == "p" else 1)
        assert np.array_equal(got, want), k


@pytest.mark.parametrize("edge_first", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_banded_pe25d_with_physics_equals_single_domain(tmp_path, world, edge_first):
    """BASELINE configs[4] on latitude bands: dynamics + solar_timestep every step.  The radiation changes theta and
    the ground temperature AFTER the post-corrector exchange; the bands radiate their ghost rows locally (the ghost
    rows of gt travel with the messages), so the next predictor's edge rows read current values without a third
    exchange per step.  Bit for bit the single domain (the oracle: matsuno_timestep, then solar_timestep)."""
    from oracle import dynamics, physics, geometry as ogeo
    hwl, steps = (12, 16, 3), 3
    mp.spawn(_worker_pe, args=(world, _free_port(), hwl, steps, str(tmp_path), edge_first, True), nprocs=world,
             join=True)
    H, W, L = hwl
    geom = ogeo.gen_geometry(H, W, L, sig_func=ogeo.manabe_sig)
    geom.heightmap[H // 2, 3] = 300.0
    st, gt, utc = _ic_pe(geom), _ic_gt(geom), UTC0
    for _ in range(steps):
        st = list(dynamics.matsuno_timestep(*st, 120.0, geom))
        st[3], gt = physics.solar_timestep(st[3], st[0], gt, 120.0, utc, geom)
        utc += 120.0
    parts = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    for k, want in zip("puvtqg", list(st) + [gt]):
        got = np.concatenate([pp[k] for pp in parts], axis=0 if k in "pg" else 1)
        assert np.array_equal(got, want), k
    # the physics did something, and a band that skipped its ghost rows would differ: theta moved
    assert not np.array_equal(st[3], dynamics.matsuno_timestep(*_ic_pe(geom), 120.0, geom)[3])


def test_rccl_unique_id_survives_transport():
    """gcmiipy_amd.rccl carries the 128-byte ncclUniqueId from rank 0 to the others as bytes: the id
    has NUL bytes inside, all 128 must arrive (no GPU, no librccl needed)"""
    import ctypes as C
    from gcmiipy_amd import rccl
    uid = rccl._UniqueId()
    raw = bytes((7 * i) % 256 if i % 5 else 0 for i in range(128))
    C.memmove(C.addressof(uid), raw, 128)
    wire = rccl.uid_to_bytes(uid)
    assert wire == raw and len(wire) == 128
    back = rccl.uid_from_bytes(wire)
    assert rccl.uid_to_bytes(back) == raw
    with pytest.raises(ValueError):
        rccl.uid_from_bytes(raw[:100])
