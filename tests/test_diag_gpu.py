"""GPU parity of the diagnostics and the flux_limiter.py drop-in: the total
variation monitor, courant_number, the fused STATS record, van_leer / calc_r / donor-cell pieces
against the golden vectors captured from the reference (g1, g2, g4)."""
import numpy as np
import pytest

from conftest import golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def g():
    import gcmiipy_amd
    assert gcmiipy_amd.device_count() >= 1, "no MI355X visible"
    return gcmiipy_amd


def test_flux_limiter_pieces_bit_exact_vs_golden(g):
    """flux_limiter.py:10-32: bit for bit, so the masks (b != 0 in calc_r, strict u > 0 in
    donor_cell_flux) are exactly the reference's; q1[3:6] are equal and u1[7] == 0 in the fixture"""
    from gcmiipy_amd import flux_limiter as fl
    d = golden("g4_tracer")
    q1, u1 = d["q1"], d["u1"]
    r = fl.calc_r(q1)
    assert np.array_equal(r, d["r1"])
    assert np.array_equal(r == 0.0, d["r1"] == 0.0) and (r[3:5] == 0.0).all()
    assert np.array_equal(fl.van_leer(r), d["phi1"])
    pts = np.asarray([fl.van_leer(x) for x in (1, 0, -2.0, 0.5, 1e30)])
    assert np.array_equal(pts, d["phi_pts"])
    assert fl.van_leer(1) == 1 and fl.van_leer(0) == 0           # flux_limiter.py:46-48
    f = fl.donor_cell_flux(q1, u1)
    assert np.array_equal(f, d["donor_flux"])
    assert f[7] == 0.0                                            # u == 0 takes ip(q) * 0
    assert np.array_equal(fl.donor_cell_advection(q1, u1, 100.0, 1.0), d["donor_adv"])
    # 100 donor-cell steps of the reference's own test setup (flux_limiter.py:73-89): conservative, bounded
    q = np.zeros(16); q[4:8] = 1.0
    u = np.full(16, 10.0)
    for _ in range(100):
        q = fl.donor_cell_advection(q, u, 100.0, 1.0)
    assert abs(q.sum() - 4.0) < 1e-12 and q.min() >= 0.0 and q.max() <= 1.0
    with pytest.raises(ValueError):
        fl.calc_r(np.zeros((3, 3)))


def test_total_variation_and_courant_vs_golden(g):
    d1, d2 = golden("g1_shifts"), golden("g2_sw2d")
    a2 = d1["a2"]
    H, W = a2.shape
    c = g.Core(g._lib.SW2D, W, H, dx=300e3)
    c.set_state(p=8000 + a2, u=a2, v=a2)
    assert abs(c.total_variation(g._lib.U) - float(d1["tv"])) < TOL * float(d1["tv"])
    assert abs(c.total_variation(g._lib.P) - float(d1["tv"])) < 1e-9 * float(d1["tv"])
    with pytest.raises(ValueError):
        c.total_variation(g._lib.T)                               # SW2D has no theta
    c.close()
    from gcmiipy_amd.matsuno_c_grid import courant_number
    got = courant_number(8000 + a2, a2, 300e3, 300.0)             # constants.py:111-112
    assert abs(got - float(d1["courant"])) < TOL * float(d1["courant"])
    got = courant_number(d2["p0"], d2["u0"], float(d2["dx"]), float(d2["dt"]))
    assert abs(got - float(d2["courant"])) < TOL * float(d2["courant"])


def test_tv_3d_axis0_is_levels_and_resident_monitor(g):
    """get_total_variation rolls axis 0: the level axis of the reference's [k][j][i] fields"""
    from gcmiipy_amd import geometry
    d = golden("g8_pe25d")
    geom = geometry.gen_geometry(24, 36, 9, sig_func=geometry.manabe_sig)
    ic = [d["dense_%s0" % k] for k in "puvtq"]
    c = g.Core(g._lib.PE25D, 36, 24, 9, geom=geom)
    c.set_state(*ic)
    for f in range(5):
        want = np.sum(np.abs(ic[f] - np.roll(ic[f], -1, 0)))
        assert abs(c.total_variation(f) - want) < TOL * want, f
    c32 = g.Core(g._lib.PE25D, 36, 24, 9, geom=geom, dtype="f32")
    c32.set_state(*ic)
    want = np.sum(np.abs(ic[1] - np.roll(ic[1], -1, 0)))
    assert abs(c32.total_variation(1) - want) < 1e-5 * want
    c32.close()
    c.close()


def test_fused_stats_equals_separate_reductions(g):
    from gcmiipy_amd import geometry
    d = golden("g8_pe25d")
    geom = geometry.gen_geometry(24, 36, 9, sig_func=geometry.manabe_sig)
    ic = [d["dense_%s0" % k] for k in "puvtq"]
    c = g.Core(g._lib.PE25D, 36, 24, 9, geom=geom)
    c.set_state(*ic)
    c.step(2, 30.0)
    area = np.ones(1)
    s = c.stats(area)
    assert s["u_max"] == c.diag(g._lib.DIAG_MAX_U) and s["u_min"] == c.diag(g._lib.DIAG_MIN_U)
    assert s["v_max"] == c.diag(g._lib.DIAG_MAX_V) and s["v_min"] == c.diag(g._lib.DIAG_MIN_V)
    assert s["ke"] == c.energy(area) and s["nans"] == 0.0
    p, u, v, t, q = c.get_state()
    assert s["u_max"] == u.max() and s["v_min"] == v.min()
    u[3, 4, 5] = np.nan
    c.set_state(p, u, v, t, q)
    s = c.stats(area)
    assert s["nans"] >= 1.0 and np.isnan(s["u_max"])
    c.close()


def test_constants_reductions_without_a_handle():
    """constants.get_total_variation / courant_number on plain arrays (gcm_array_stats) vs G1's `tv` /
    `courant`, G2's `courant`, and NumPy on other shapes; the constants themselves vs the oracle's"""
    from gcmiipy_amd import constants as c
    from oracle import constants as oc, grid
    for k in ("Rd", "Cp", "kappa", "P0", "standard_pressure", "standard_temperature", "G", "radius", "mu_air", "Rv"):
        assert getattr(c, k) == getattr(oc, k), k
    d2 = golden("g2_sw2d")
    assert abs(c.courant_number(d2["p0"], d2["u0"], float(d2["dx"]), float(d2["dt"])) / float(d2["courant"]) - 1) < 1e-12
    rng = np.random.default_rng(5)
    for shape in ((11,), (5, 7), (3, 5, 7), (1, 9), (64, 130)):
        q = rng.standard_normal(shape)
        assert abs(c.get_total_variation(q) / grid.get_total_variation(q) - 1) < 1e-12 if q.shape[0] > 1 else c.get_total_variation(q) == 0.0
    d1 = golden("g1_shifts")
    assert abs(c.get_total_variation(d1["a2"]) / float(d1["tv"]) - 1) < 1e-12
    assert abs(c.courant_number(8000 + d1["a2"], d1["a2"], 300e3, 300.0) / float(d1["courant"]) - 1) < 1e-12


def test_courant_number_propagates_nan():
    """np.max(u) is NaN when u holds a NaN (constants.py:111-112): the stability monitor must not hide a
    blown-up field behind fmax"""
    from gcmiipy_amd import constants as c
    rng = np.random.default_rng(6)
    u = rng.standard_normal((33, 70))
    p = 8000 + rng.standard_normal((33, 70))
    assert np.isfinite(c.courant_number(p, u, 300e3, 300.0))
    u[17, 3] = np.nan
    assert np.isnan(c.courant_number(p, u, 300e3, 300.0))
    assert np.isnan(c.get_total_variation(u))


def test_band_total_variation_sums_to_global_and_needs_current_ghosts(g):
    """GCM_DIAG_TV_* on latitude bands: the last row is differenced against the south ghost row, so the
    band partials add up to the single domain's figure once the CURRENT state's ghost rows have been
    exchanged -- and the call refuses (GCM_ERR_STATE) while they are stale, i.e. right after a step"""
    import torch
    from gcmiipy_amd.bands import split_rows
    from test_bands_gpu import _exchange, _ic2d
    H, W, nb = 37, 130, 3
    f = _ic2d((H, W))
    kw = dict(dx=300e3, tracer=g._lib.TRACER_VANLEER)
    ref = g.Core(g._lib.SW2D_TEMP, W, H, **kw)
    ref.set_state(**f)
    cores = []
    for r, (row0, n) in enumerate(split_rows(H, nb)):
        c = g.Core(g._lib.SW2D_TEMP, W, n, nranks=nb, rank=r, global_height=H, row0=row0, **kw)
        c.set_state(**{k: v[row0:row0 + n] for k, v in f.items()})
        cores.append(c)
    with pytest.raises(g.GcmError):
        cores[0].total_variation(g._lib.P)            # fresh state: ghost rows never filled
    for rnd in range(2):
        _exchange(cores, torch)
        for fld in (g._lib.P, g._lib.U, g._lib.Q):
            want = ref.total_variation(fld)
            got = sum(c.total_variation(fld) for c in cores)
            assert abs(got / want - 1) < 1e-12, (rnd, fld, got, want)
        ref.step(1, 300.0)
        for c in cores:
            c.step_interior(300.0)
        for c in cores:
            c.step_boundary(300.0)
        with pytest.raises(g.GcmError):
            cores[1].total_variation(g._lib.U)        # stepped: the ghost rows belong to the old state
    ref.close()
    for c in cores:
        c.close()
