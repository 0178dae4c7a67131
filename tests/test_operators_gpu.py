"""GPU parity of the per-operator drop-ins (the functions matsuno_c_grid.py, viscosity.py,
matsumo_temp.py and temperature.py export, behind gcm_sw2d_op) vs the golden vectors G2 / G3 / G11
captured from the reference, the reference's own known-answer tests (test_matsumo.py, test_viscosity.py)
and the oracle on ragged shapes."""
import numpy as np
import pytest

from conftest import golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_matsuno_c_grid_operators_vs_golden():
    from gcmiipy_amd import matsuno_c_grid as m
    d = golden("g2_sw2d")
    u, v, p, dx = d["u0"], d["v0"], d["p0"], float(d["dx"])
    assert rel_err(m.advection_of_velocity_u(u, v, dx), d["adv_u"]) < TOL
    assert rel_err(m.advection_of_velocity_v(u, v, dx), d["adv_v"]) < TOL
    assert rel_err(m.geopotential_gradient_u(p, dx), d["ggu"]) < TOL
    assert rel_err(m.geopotential_gradient_v(p, dx), d["ggv"]) < TOL
    assert rel_err(m.advection_of_geopotential(u, v, p, dx), d["adv_p"]) < TOL


def test_matsumo_temp_and_viscosity_operators_vs_golden():
    from gcmiipy_amd import matsumo_temp as mt, viscosity
    d = golden("g3_sw2d_temp")
    u, p, t, dx = d["u0"], d["p0"], d["t0"], float(d["dx"])
    rho = mt.density_from(p, t)
    assert rel_err(rho, d["density"]) < TOL
    assert rel_err(mt.geopotential_from(rho, p), d["geo"]) < TOL
    sc = mt.scaling(p, t, dx)
    assert rel_err(sc, d["scaled"]) < TOL
    assert rel_err(mt.unscaling(p, sc, dx), d["unscaled"]) < TOL
    assert rel_err(viscosity.finite_laplacian_2d(u, dx), d["lap_u"]) < TOL
    assert rel_err(viscosity.incompressible_viscosity_2d(u, float(d["mu_air"]), dx), d["visc_u"]) < TOL


def test_temperature_conversions_vs_golden():
    from gcmiipy_amd import temperature as tm
    d = golden("g11_temperature")
    theta = tm.to_potential_temp(d["tt"], d["p"])
    assert rel_err(theta, d["theta"]) < TOL
    assert rel_err(tm.to_true_temp(d["theta"], d["p"]), d["back"]) < TOL
    assert rel_err(tm.to_density(d["tt"], d["p"]), d["rho"]) < TOL
    assert abs(tm.to_true_temp(tm.to_potential_temp(273.16, 101325.0), 101325.0) - 273.16) < 1e-9   # temperature.py:33-43
    with pytest.raises(ValueError):
        tm.to_true_temp(d["tt"], d["p"][:, :-1])                     # the reference asserts equal shapes


def test_reference_known_answers():
    from gcmiipy_amd import matsuno_c_grid as m, viscosity
    p = np.full((3, 3), 1.0)
    p[1, 1] = 0
    assert m.geopotential_gradient_v(p, 1.0)[1, 1] == 9.8                             # test_matsumo.py:24-30
    u = np.zeros((5, 5))
    u[2, 2] = 1.0
    lap = viscosity.finite_laplacian_2d(u, 2.0)
    want = np.zeros((5, 5))
    want[2, 2], want[1, 2], want[3, 2], want[2, 1], want[2, 3] = -1.0, 0.25, 0.25, 0.25, 0.25
    assert np.array_equal(lap, want)


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (5, 1), (3, 130), (67, 9)])
def test_operators_vs_oracle_ragged(shape):
    from gcmiipy_amd import matsuno_c_grid as m, matsumo_temp as mt, viscosity
    from oracle import sw2d, sw2d_temp
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    u, v = rng.standard_normal(shape), rng.standard_normal(shape)
    p = 101325 + rng.standard_normal(shape)
    t = 273.16 + rng.standard_normal(shape)
    dx, dt = 2.5e5, 200.0
    for got, want in ((m.advection_of_velocity_u(u, v, dx), sw2d.advection_of_velocity_u(u, v, dx)),
                      (m.advection_of_velocity_v(u, v, dx), sw2d.advection_of_velocity_v(u, v, dx)),
                      (m.geopotential_gradient_u(p, dx), sw2d.geopotential_gradient_u(p, dx)),
                      (m.geopotential_gradient_v(p, dx), sw2d.geopotential_gradient_v(p, dx)),
                      (m.advection_of_geopotential(u, v, p, dx), sw2d.advection_of_geopotential(u, v, p, dx)),
                      (viscosity.incompressible_viscosity_2d(u, 1.8e-5, dx), sw2d_temp.incompressible_viscosity_2d(u, 1.8e-5, dx))):
        assert rel_err(got, want) < TOL
    # advect_t = scaling -> advection_of_geopotential -> unscaling (matsumo_temp.py:38-42)
    pb = p + rng.standard_normal(shape)
    sc = sw2d_temp.scaling(p, t, dx)
    want = sw2d_temp.unscaling(pb, sc - dt * sw2d.advection_of_geopotential(u, v, sc, dx), dx)
    assert rel_err(mt.advect_t(t, u, v, p, pb, dx, dt), want) < TOL


def test_no_limits_2d_operators_vs_golden():
    """the 2-D primitive-equation operators one by one (no_limits_2d.py:21-101) vs G10"""
    from gcmiipy_amd import no_limits_2d as n2
    d = golden("g10_pe2d")
    p, u, v, t, dx = d["p0"], d["u0"], d["v0"], d["t0"], float(d["dx"])
    pu, pv = n2.calc_pu(p, u), n2.calc_pv(p, v)
    assert rel_err(pu, d["pu"]) < TOL and rel_err(pv, d["pv"]) < TOL
    assert rel_err(n2.un_pu(d["pu"], p), u) < TOL and rel_err(n2.un_pv(d["pv"], p), v) < TOL
    assert rel_err(n2.advec_p(d["pu"], d["pv"], dx), d["advec_p"]) < TOL
    dut, dvt = n2.advec_m(p, u, v, dx)
    assert rel_err(dut, d["dut"]) < TOL and rel_err(dvt, d["dvt"]) < TOL
    pgu, pgv = n2.pgf(p, t, dx)
    assert rel_err(pgu, d["pgu"]) < TOL and rel_err(pgv, d["pgv"]) < TOL
    assert rel_err(n2.advec_t(d["pu"], d["pv"], t, dx), d["advec_t"]) < TOL
