"""The oracle (numpy restatement) must equal the reference BIT FOR BIT on the
golden vectors that tests/golden/make_golden.py produced by running the
unmodified reference source."""
import numpy as np
import pytest

from conftest import golden
from oracle import (grid, sw2d, sw2d_temp, tracer, pe2d, oned, geometry, lowpass,
                    dynamics, driver, temperature, constants)


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), float(np.nanmax(np.abs(a - b)))


def test_g1_shifts():
    d = golden("g1_shifts")
    a1, a2, a3 = d["a1"], d["a2"], d["a3"]
    for n in ("ipj", "imj", "ijp", "ijm", "imjp", "iph", "imh", "jph", "jmh"):
        f = getattr(grid, n)
        same(f(a2), d["c2_" + n])
        same(f(a3), d["c3_" + n])
        same(f(a2), d["c3on2_" + n])
    for n in ("kp", "km", "kph", "kmh"):
        same(getattr(grid, n)(a3), d["c3_" + n])
    same(grid.gradi(a2, 3.0), d["c2_gradi"])
    same(grid.gradj(a2, 3.0), d["c2_gradj"])
    same(grid.gradi(a3, 3.0), d["c3_gradi"])
    same(grid.gradj(a3, 3.0), d["c3_gradj"])
    same(grid.ip(a1), d["c1_ip"])
    same(grid.im(a1), d["c1_im"])
    same(grid.iph1(a1), d["c1_iph"])
    same(grid.imh1(a1), d["c1_imh"])
    same(grid.div(a1, 3.0), d["c1_div"])
    same(grid.divu(a1, 3.0), d["c1_divu"])
    same(grid.gradh(a1, 3.0), d["c1_gradh"])
    same(grid.get_total_variation(a2), d["tv"])
    same(grid.courant_number(8000 + a2, a2, 300e3, 300.0), d["courant"])


def test_g2_sw2d_operators_and_steps():
    d = golden("g2_sw2d")
    u, v, p, dx, dt = d["u0"], d["v0"], d["p0"], float(d["dx"]), float(d["dt"])
    same(sw2d.advection_of_velocity_u(u, v, dx), d["adv_u"])
    same(sw2d.advection_of_velocity_v(u, v, dx), d["adv_v"])
    same(sw2d.geopotential_gradient_u(p, dx), d["ggu"])
    same(sw2d.geopotential_gradient_v(p, dx), d["ggv"])
    same(sw2d.advection_of_geopotential(u, v, p, dx), d["adv_p"])
    same(grid.courant_number(p, u, dx, dt), d["courant"])
    for n in range(1, 11):
        u, v, p = sw2d.matsumo_scheme(u, v, p, dx, dt)
        if n in (1, 2, 10):
            same(u, d["u%d" % n]); same(v, d["v%d" % n]); same(p, d["p%d" % n])


def test_g2_reference_main_ic():
    """matsuno_c_grid.py:145-158 IC, 10 steps; also the three scalars SURVEY.md
    8c recorded from the survey's own run of the reference."""
    d = golden("g2_sw2d")
    side = 64
    u = np.zeros((side, side)); v = np.zeros((side, side)); p = np.full((side, side), 8000.0)
    u[32, 32] += 30
    for _ in range(10):
        u, v, p = sw2d.matsumo_scheme(u, v, p, 300 * 1000.0, 300.0)
    same(u, d["main_u10"]); same(v, d["main_v10"]); same(p, d["main_p10"])
    assert np.max(np.abs(u)) == 14.818898821091882
    assert p[32, 32] == 7991.470895298706
    assert p.sum() == 32768000.0


def test_g3_sw2d_temp():
    d = golden("g3_sw2d_temp")
    u, v, p, t, dx, dt = d["u0"], d["v0"], d["p0"], d["t0"], float(d["dx"]), float(d["dt"])
    assert constants.mu_air == float(d["mu_air"])
    rho = sw2d_temp.density_from(p, t)
    same(rho, d["density"])
    same(sw2d_temp.geopotential_from(rho, p), d["geo"])
    same(sw2d_temp.scaling(p, t, dx), d["scaled"])
    same(sw2d_temp.unscaling(p, sw2d_temp.scaling(p, t, dx), dx), d["unscaled"])
    same(sw2d_temp.finite_laplacian_2d(u, dx), d["lap_u"])
    same(sw2d_temp.incompressible_viscosity_2d(u, constants.mu_air, dx), d["visc_u"])
    for n in range(1, 6):
        u, v, p, t = sw2d_temp.matsumo_scheme(u, v, p, t, dx, dt)
        if n in (1, 5):
            for k, x in zip("uvpt", (u, v, p, t)):
                same(x, d["%s%d" % (k, n)])


def test_g4_tracer():
    d = golden("g4_tracer")
    V, q, p, t, dt = d["V0"], d["q0"], d["p0"], d["t0"], float(d["dt"])
    sc = tuple(float(x) for x in d["sc"])
    for ax in (0, 1):
        same(tracer.upwind_axis(dt, sc, V, q, ax), d["upwind_axis%d" % ax])
        same(tracer.upwind_axis_finite(dt, sc, V, q, ax), d["upwind_axis_finite%d" % ax])
        same(tracer.fv_advect_axis_upwind(dt, sc, V, q, ax), d["fv_upwind%d" % ax])
        same(tracer.fv_advect_axis_upwind_finite(dt, sc, V, q, ax), d["fv_upwind_finite%d" % ax])
        same(tracer.fv_advect_axis_plain(dt, sc, V, q, ax), d["fv_plain%d" % ax])
        same(tracer.fv_advect_axis_plain_finite(dt, sc, V, q, ax), d["fv_plain_finite%d" % ax])
        same(tracer.pgf_c_grid_axis(p, sc, ax), d["pgf_axis%d" % ax])
    same(tracer.corner_transport_2d(dt, sc, V, q), d["ctu"])
    same(tracer.finite_volume_advection(dt, sc, V, q), d["fva"])
    same(tracer.pgf_c_grid(dt, sc, p, t), d["pgf_c_grid"])
    same(tracer.pgf_templess(dt, sc, p), d["pgf_templess"])
    same(tracer.pressure_at_edge(p), d["p_edge"])
    same(tracer.pressure_at_edge_one_d(p), d["p_edge_1d"])
    same(tracer.advect_with_momentum(dt, sc, V, p), d["adv_mom"])
    same(tracer.pgf_one_d(dt, sc[0], p), d["pgf_one_d"])
    same(tracer.pgf_one_d(dt, sc[1], p, 1), d["pgf_one_d_axis1"])
    same(tracer.pgf_one_d(dt, sc[0], p[:, 0]), d["pgf_one_d_line"])
    for ax in (0, 1):
        same(tracer.gradient(p, sc, ax), d["gradient%d" % ax])
    same(tracer.pressure_gradient(dt, sc, p, t), d["pressure_gradient"])
    q1, u1 = d["q1"], d["u1"]
    r = tracer.calc_r(q1)
    same(r, d["r1"])
    assert (r[[3, 4]] == 0).all()          # zero-denominator rule (flux_limiter.py:19)
    same(tracer.van_leer(r), d["phi1"])
    same([tracer.van_leer(x) for x in (1, 0, -2.0, 0.5, 1e30)], d["phi_pts"])
    same(tracer.donor_cell_flux(q1, u1), d["donor_flux"])
    same(tracer.donor_cell_advection(q1, u1, 100.0, 1.0), d["donor_adv"])


def test_g4_state_dict_driver():
    """test_2d.py:240-252 setup through the headless run_2d_with_ft."""
    d = golden("g4_tracer")
    V = np.zeros((2, 4, 4)); q = np.zeros((4, 4))
    q[1:2, 1:2] = 1.0
    V[0][:] = 2.0; V[1][:] = -2.0

    def ft(V, q):
        return {"V": V, "q": tracer.corner_transport_2d(1.0, (10.0, 10.0), V, q)}

    ok, final, tv = driver.run_2d_with_ft({"V": V, "q": q}, ft)
    assert ok is True
    same(final["q"], d["runfunc_q400"])
    same(tv, d["runfunc_tv"])
    qq = q
    for _ in range(400):
        qq = tracer.finite_volume_advection(1.0, (10.0, 10.0), V, qq)
    same(qq, d["fv_q400"])


GEOMS = [(24, 36, 9), (8, 16, 4), (12, 20, 5)]


@pytest.mark.parametrize("hwl", GEOMS)
@pytest.mark.parametrize("sname", ["manabe_sig", "equal_sig"])
def test_g5_geometry(hwl, sname):
    d = golden("g5_geometry")
    h, w, l = hwl
    g = geometry.gen_geometry(h, w, l, sig_func=getattr(geometry, sname))
    pre = "g_%d_%d_%d_%s_" % (h, w, l, sname)
    for k in ("sige", "sigt", "sigb", "dsig", "sig", "dsigv", "dx_j", "dx_h", "dy", "ptop",
              "heightmap", "area", "lat", "long"):
        same(getattr(g, k), d[pre + k])


def test_g5_geometry_large_and_square():
    d = golden("g5_geometry")
    g = geometry.gen_geometry(720, 1440, 24, sig_func=geometry.manabe_sig)
    for k in ("sige", "sigt", "sigb", "dsig", "sig", "dsigv", "dx_j", "dx_h", "dy", "ptop",
              "area", "lat", "long"):
        same(getattr(g, k), d["g_720_1440_24_manabe_sig_" + k])
    g = geometry.gen_square_geometry(6, 10, 3, 1000.0, 1200.0)
    for k in ("sige", "sigt", "sigb", "dsig", "sig", "dsigv", "dx_j", "dx_h", "dy", "ptop",
              "heightmap"):
        same(getattr(g, k), d["sq_6_10_3_" + k])


@pytest.mark.parametrize("lhw", [(3, 8, 16), (9, 24, 36), (2, 6, 10)])
def test_g6_lowpass(lhw):
    d = golden("g6_lowpass")
    l, h, w = lhw
    g = geometry.gen_geometry(h, w, l)
    same(lowpass.arakawa_1977(d["in_%d_%d_%d" % lhw], g), d["out_%d_%d_%d" % lhw])


def test_g7_half_step_intermediates():
    d = golden("g7_half_step")
    L, H, W = d["u0"].shape
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    geom.heightmap[...] = d["heightmap"]
    base = tuple(d[k + "0"] for k in "puvtq")
    dt = float(d["dt"])
    stage = base
    for tag in ("pred", "corr"):
        tap = {}
        nxt = dynamics.half_timestep(*base, *stage, dt, geom, _tap=tap)
        for k in ("spu", "spv", "pit", "sd", "dut", "dvt", "pgu", "pgv", "phiu", "phiv",
                  "dus", "dvs", "pgfu"):
            same(tap[k], d["%s_%s" % (tag, k)])
        same(dynamics.compute_geopotential(stage[0], stage[3], geom), d[tag + "_phi"])
        same(dynamics.advec_t(tap["spu"], tap["spv"], stage[3], geom), d[tag + "_advec_t"])
        same(dynamics.advec_sig(tap["sd"], stage[3], geom), d[tag + "_advec_sig_t"])
        for k, x in zip(("p_n", "u_n", "v_n", "t_n", "q_n"), nxt):
            same(x, d["%s_%s" % (tag, k)])
        stage = nxt
    for k, x in zip("puvtq", dynamics.matsuno_timestep(*base, dt, geom)):
        same(x, d["step_" + k])


def test_g8_harness_and_dense():
    d = golden("g8_pe25d")
    # run_model at its main() size with STATS
    stats = {k: [] for k in ("u_max", "u_min", "v_max", "v_min", "ke")}
    snaps = {}

    def cb(p, u, v, t, q, _n=[0]):
        _n[0] += 1
        if _n[0] in (1, 3, 10):
            snaps[_n[0]] = (p, u, v, t, q)

    driver.run_model(8, 8, 3, 1800.0, 10, cb, stats=stats)
    for n, st in snaps.items():
        for k, x in zip("puvtq", st):
            same(x, d["harness%d_%s" % (n, k)])
    for k in ("u_max", "u_min", "v_max", "v_min"):
        same(stats[k], d["harness_stats_" + k])
    same(np.asarray(stats["ke"]), d["harness_stats_ke"])
    # initial conditions
    for (h, w, l) in ((24, 36, 9), (8, 16, 4)):
        g = geometry.gen_geometry(h, w, l, sig_func=geometry.manabe_sig)
        p, u, v, t, q, gr = driver.gen_initial_conditions(g)
        for k, x in zip(("p", "u", "v", "t", "q", "gt"), (p, u, v, t, q, gr.gt)):
            same(x, d["ic_%d_%d_%d_%s" % (h, w, l, k)])
    # harness IC at 36x24x9
    geom = geometry.gen_geometry(24, 36, 9, sig_func=geometry.manabe_sig)
    p, u, v, t, q, _ = driver.gen_initial_conditions(geom)
    v[0, 0, 0] = 0.1
    u *= 0
    st = (p, u, v, t, q)
    for n in range(1, 11):
        st = dynamics.matsuno_timestep(*st, 900.0, geom)
        if n in (1, 3, 10):
            for k, x in zip("puvtq", st):
                same(x, d["h36_%d_%s" % (n, k)])
    # dense random IC
    st = tuple(d["dense_%s0" % k] for k in "puvtq")
    for n in range(1, 11):
        st = dynamics.matsuno_timestep(*st, float(d["dense_dt"]), geom)
        if n in (1, 3, 10):
            for k, x in zip("puvtq", st):
                same(x, d["dense%d_%s" % (n, k)])


def test_g8_geography_bump():
    """test_geography.py:6-23,49: H=1, W=16, L=17, heightmap[0,8] = 1000 m."""
    d = golden("g8_pe25d")
    geom = geometry.gen_geometry(1, 16, 17, sig_func=geometry.manabe_sig)
    p, u, v, t, q, gr = driver.gen_initial_conditions(geom)
    v[0, 0, 0] = 0.1
    u *= 0
    geom.heightmap[0, 8] = 1000
    st = (p, u, v, t, q)
    for _ in range(3):
        st = dynamics.matsuno_timestep(*st, 1800.0, geom)
    for k, x in zip("puvtq", st):
        same(x, d["bump3_" + k])
    same(driver.calc_energy(*st, gr, geom), d["bump3_energy"])


def test_g9_oned():
    d = golden("g9_oned")
    st = tuple(d[k + "0"] for k in "putq")
    s2 = st
    for _ in range(2):
        s2 = oned.matsuno_timestep(*s2, 900.0, 70000.0)
    for k, x in zip("putq", s2):
        same(x, d[k + "2_900"])
    for _ in range(10):
        st = oned.matsuno_timestep(*st, float(d["dt"]), float(d["dx"]))
    for k, x in zip("putq", st):
        same(x, d[k + "10"])
    p, u, t, q = (d["op_" + k] for k in "putq")             # the operators one by one, no_limits.py:50-112
    pu = oned.calc_pu(u, p)
    same(pu, d["op_calc_pu"])
    same(oned.un_pu(pu, p), d["op_un_pu"])
    same(oned.advec_q(u, q, 70000.0), d["op_advec_q"])
    same(oned.advec_p(pu, 70000.0), d["op_advec_p"])
    same(oned.advec_pu(p, pu, u, 70000.0), d["op_advec_pu"])
    same(oned.advec_t(pu, t, 70000.0), d["op_advec_t"])
    same(oned.pgf(p, t, 70000.0), d["op_pgf"])
    qq = d["adv_q0"]
    V = np.full((1, 161), 2.0)
    for _ in range(400):
        qq = tracer.fv_advect_axis_upwind(1.0, (10.0,), V, qq, 0)
    same(qq, d["adv_q400"])


def test_g10_pe2d():
    d = golden("g10_pe2d")
    p, u, v, t, q = (d[k + "0"] for k in "puvtq")
    dx, dt = float(d["dx"]), float(d["dt"])
    pu, pv = pe2d.calc_pu(p, u), pe2d.calc_pv(p, v)
    same(pu, d["pu"]); same(pv, d["pv"])
    same(pe2d.advec_p(pu, pv, dx), d["advec_p"])
    dut, dvt = pe2d.advec_m(p, u, v, dx)
    same(dut, d["dut"]); same(dvt, d["dvt"])
    pgu, pgv = pe2d.pgf(p, t, dx)
    same(pgu, d["pgu"]); same(pgv, d["pgv"])
    same(pe2d.advec_t(pu, pv, t, dx), d["advec_t"])
    st = (p, u, v, t, q)
    for n in range(1, 6):
        st = pe2d.matsuno_timestep(*st, dt, dx)
        if n in (1, 5):
            for k, x in zip("puvtq", st):
                same(x, d["%s%d" % (k, n)])


def test_g11_temperature_and_constants():
    d = golden("g11_temperature")
    th = temperature.to_potential_temp(d["tt"], d["p"])
    same(th, d["theta"])
    same(temperature.to_true_temp(th, d["p"]), d["back"])
    same(temperature.to_density(d["tt"], d["p"]), d["rho"])
    for k in ("kappa", "P0", "G", "Rd", "Cp", "radius", "Rv"):
        assert getattr(constants, k) == float(d[k]), k


def test_g12_coriolis_branch():
    """the disabled Coriolis branch (dynamics.py:83-92), reference run with `if False` flipped in
    memory by make_golden.g12_coriolis"""
    d = golden("g12_coriolis")
    L, H, W = d["u0"].shape
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    st = tuple(d[k + "0"] for k in "puvtq")
    dut, dvt = dynamics.advec_m_pu(st[0], st[1], st[2], dynamics.calc_pu(st[0], st[1]),
                                   dynamics.calc_pv(st[0], st[2]), geom, coriolis=True)
    same(dut, d["dut"]); same(dvt, d["dvt"])
    for n in (1, 2):
        st = dynamics.matsuno_timestep(*st, float(d["dt"]), geom, coriolis=True)
        for k, x in zip("puvtq", st):
            same(x, d["step%d_%s" % (n, k)])


def test_g13_radiation():
    from oracle import physics
    d = golden("g13_radiation")
    L, H, W = d["t0"].shape
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    p, t, gt = d["p0"], d["t0"], d["gt0"]
    for tag in "ab":
        utc = float(d["utc_" + tag])
        same(physics.zenith_angle(geom.long, geom.lat, utc, geom), d["sza_" + tag])
        tp = p * geom.sig + geom.ptop
        tt = temperature.to_true_temp(t, tp)
        dTdt, dtg = physics.basic_grey_radiation(p, tp, tt, gt, 0.1, 0.9, 0.3, utc, geom)
        same(dTdt, d["dTdt_" + tag]); same(dtg, d["dtg_" + tag])
        t_n, gt_n = physics.solar_timestep(t, p, gt, float(d["dt"]), utc, geom)
        same(t_n, d["t_n_" + tag]); same(gt_n, d["gt_n_" + tag])
