"""The reference's own known-answer tests for the hot path, restated against
the oracle: these are the only numbers the
reference's test-suite itself pins."""
import numpy as np

from oracle import grid, sw2d, tracer, temperature, constants


def _p3():
    p = np.full((3, 3), 1.0)
    p[1, 1] = 0
    return p


def test_ipj_direction():                       # test_matsumo.py:9-14
    assert grid.ipj(_p3())[1, 0] == 0


def test_ijp_direction():                       # test_matsumo.py:16-21
    assert grid.ijp(_p3())[0, 1] == 0


def test_pgf_v_equals_G():                      # test_matsumo.py:24-29
    assert sw2d.geopotential_gradient_v(_p3(), 1.0)[1, 1] == constants.G


def test_pressure_at_edges():                   # test_2d.py:176-181
    assert tracer.pressure_at_edge(_p3())[0][0][1] == 0.5


def test_van_leer_points():                     # flux_limiter.py:46-48
    assert tracer.van_leer(1) == 1
    assert tracer.van_leer(0) == 0


def test_temperature_round_trip():              # temperature.py:31-41
    tt, p = constants.standard_temperature, constants.standard_pressure
    t = temperature.to_potential_temp(tt, p)
    assert abs(temperature.to_true_temp(t, p) - tt) < 1e-7   # assertAlmostEqual


def _tv_ic():                                   # test_2d.py:30-44, side_length 4
    V = np.zeros((2, 4, 4)); q = np.zeros((4, 4))
    q[1:2, 1:2] = 1.0
    V[0][:] = 2.0; V[1][:] = -2.0
    return V, q


def test_ctu_total_variation_bounded():         # test_2d.py:47-80
    V, q = _tv_ic()
    tv0 = grid.get_total_variation(q)
    for _ in range(400):
        q = tracer.corner_transport_2d(1.0, (10.0, 10.0), V, q)
        assert grid.get_total_variation(q) <= tv0 + 0.00001


def test_fv_total_variation_bounded():          # test_2d.py:140-173
    V, q = _tv_ic()
    tv0 = grid.get_total_variation(q)
    for _ in range(400):
        q = tracer.finite_volume_advection(1.0, (10.0, 10.0), V, q)
        assert grid.get_total_variation(q) <= tv0 + 0.00001


def test_limited_tracer_properties():
    """Build's own van-Leer composition (parity UNPINNED):
    reduces to the reference upwind step when phi == 0, conserves mass, and is
    TV-bounded by the reference's own criterion (test_2d.py:69-72)."""
    rng = np.random.default_rng(0)
    V = rng.standard_normal((2, 8, 12)); q = rng.random((8, 12))
    a = tracer.limited_advection(0.5, (10.0, 12.0), V, q, limiter=False)
    b = tracer.finite_volume_advection(0.5, (10.0, 12.0), V, q)
    assert np.array_equal(a, b)
    V, q = _tv_ic()
    tv0 = grid.get_total_variation(q)
    m0 = q.sum()
    for _ in range(400):
        q = tracer.limited_advection(1.0, (10.0, 10.0), V, q)
        assert grid.get_total_variation(q) <= tv0 + 0.00001
        assert abs(q.sum() - m0) < 1e-12
    assert q.min() > -1e-12
