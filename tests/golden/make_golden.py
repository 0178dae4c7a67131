#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE UNMODIFIED REFERENCE SOURCE.

Build-container only (needs /root/reference; never runs on the GPU box).  The
image has no `pint`, so the unit-bookkeeping stand-in in ./_pint_standin is put
first on sys.path; `np.float` (removed in NumPy 1.24, used by the reference)
is aliased to float.  Everything numeric is executed by the reference's own
functions; this script only builds seeded inputs, strips units (`.m`) and
saves.  Inputs are SI (m, s, Pa, K, kg), so no unit-conversion factor other
than 1.0 enters except the literals listed in oracle/__init__.py.

Usage:  python tests/golden/make_golden.py        (rewrites the .npz files)
"""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_pint_standin"))
sys.path.insert(1, REF)

import numpy as np  # noqa: E402

np.float = float  # reference predates NumPy 1.24

_sink = io.StringIO()
with contextlib.redirect_stdout(_sink):
    import constants as C  # noqa: E402
    import coordinates as c2  # noqa: E402
    import coordinates_1d as c1  # noqa: E402
    import coordinates_3d as c3  # noqa: E402
    import dynamics  # noqa: E402
    import flux_limiter  # noqa: E402
    import geometry  # noqa: E402
    import low_pass  # noqa: E402
    import matsumo_temp  # noqa: E402
    import matsuno_c_grid  # noqa: E402
    import no_limits  # noqa: E402
    import no_limits_2_5d  # noqa: E402
    import no_limits_2d  # noqa: E402
    import temperature  # noqa: E402
    import two_d  # noqa: E402
    import viscosity  # noqa: E402

U = C.units
MS = U.m / U.s


def m(x):
    """strip the stand-in wrapper, recursively"""
    if isinstance(x, (tuple, list)):
        return [m(i) for i in x]
    return np.asarray(getattr(x, "m", x))


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(_sink):
        return f(*a, **k)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("%-22s %4d arrays %8.1f KB" % (name, len(arrs), os.path.getsize(path) / 1024))


# --------------------------------------------------------------------- G1
def g1_shifts():
    rng = np.random.default_rng(1)
    a2 = rng.standard_normal((5, 7))
    a3 = rng.standard_normal((3, 5, 7))
    a1 = rng.standard_normal(11)
    out = dict(a1=a1, a2=a2, a3=a3)
    for n in ("ipj", "imj", "ijp", "ijm", "imjp", "iph", "imh", "jph", "jmh"):
        out["c2_" + n] = m(getattr(c2, n)(a2 * U.m))
        out["c3_" + n] = m(getattr(c3, n)(a3 * U.m))
        out["c3on2_" + n] = m(getattr(c3, n)(a2 * U.m))
    for n in ("kp", "km", "kph", "kmh"):
        out["c3_" + n] = m(getattr(c3, n)(a3 * U.m))
    out["c2_gradi"] = m(c2.gradi(a2 * U.m, 3.0 * U.m))
    out["c2_gradj"] = m(c2.gradj(a2 * U.m, 3.0 * U.m))
    out["c3_gradi"] = m(c3.gradi(a3 * U.m, 3.0 * U.m))
    out["c3_gradj"] = m(c3.gradj(a3 * U.m, 3.0 * U.m))
    for n in ("ip", "im", "iph", "imh"):
        out["c1_" + n] = m(getattr(c1, n)(a1 * U.m))
    out["c1_div"] = m(c1.div(a1 * U.m, 3.0 * U.m))
    out["c1_divu"] = m(c1.divu(a1 * U.m, 3.0 * U.m))
    out["c1_gradh"] = m(c1.gradh(a1 * U.m, 3.0 * U.m))
    out["tv"] = m(C.get_total_variation(a2 * U.m))
    out["courant"] = m(C.courant_number((8000 + a2) * U.m, a2 * MS, 300e3 * U.m, 300 * U.s))
    save("g1_shifts", **out)


# --------------------------------------------------------------------- G2
def g2_sw2d():
    rng = np.random.default_rng(2)
    shape = (16, 32)
    u0 = rng.standard_normal(shape)
    v0 = rng.standard_normal(shape)
    p0 = 8000 + rng.standard_normal(shape)
    dx, dt = 300e3, 300.0
    u, v, p = u0 * MS, v0 * MS, p0 * U.m
    DX, DT = dx * U.m, dt * U.s
    out = dict(u0=u0, v0=v0, p0=p0, dx=dx, dt=dt)
    out["adv_u"] = m(matsuno_c_grid.advection_of_velocity_u(u, v, DX))
    out["adv_v"] = m(matsuno_c_grid.advection_of_velocity_v(u, v, DX))
    out["ggu"] = m(matsuno_c_grid.geopotential_gradient_u(p, DX))
    out["ggv"] = m(matsuno_c_grid.geopotential_gradient_v(p, DX))
    out["adv_p"] = m(matsuno_c_grid.advection_of_geopotential(u, v, p, DX))
    out["courant"] = m(matsuno_c_grid.courant_number(p, u, DX, DT))
    for n in range(1, 11):
        u, v, p = matsuno_c_grid.matsumo_scheme(u, v, p, DX, DT)
        if n in (1, 2, 10):
            out["u%d" % n], out["v%d" % n], out["p%d" % n] = m(u), m(v), m(p)
    # the reference main() initial condition (matsuno_c_grid.py:145-158), 10 steps,
    # with dx given in km as the reference does
    side = 64
    u = np.zeros((side, side)) * MS
    v = np.zeros((side, side)) * MS
    p = np.zeros((side, side)) * U.m
    p[:] = 8000 * U.m
    u[side // 2, side // 2] += 30 * MS
    for _ in range(10):
        u, v, p = matsuno_c_grid.matsumo_scheme(u, v, p, 300 * U.km, 300 * U.s)
    out["main_u10"], out["main_v10"], out["main_p10"] = m(u), m(v), m(p)
    save("g2_sw2d", **out)


# --------------------------------------------------------------------- G3
def g3_sw2d_temp():
    rng = np.random.default_rng(3)
    shape = (16, 32)
    u0 = rng.standard_normal(shape)
    v0 = rng.standard_normal(shape)
    p0 = 101325 + rng.standard_normal(shape)
    t0 = 273.16 + rng.standard_normal(shape)
    dx, dt = 300e3, 300.0
    u, v, p, t = u0 * MS, v0 * MS, p0 * U.Pa, t0 * U.K
    DX, DT = dx * U.m, dt * U.s
    out = dict(u0=u0, v0=v0, p0=p0, t0=t0, dx=dx, dt=dt)
    rho = matsumo_temp.density_from(p, t)
    out["density"] = m(rho)
    out["geo"] = m(matsumo_temp.geopotential_from(rho, p))
    out["scaled"] = m(matsumo_temp.scaling(p, t, DX))
    out["unscaled"] = m(matsumo_temp.unscaling(p, matsumo_temp.scaling(p, t, DX), DX))
    out["lap_u"] = m(viscosity.finite_laplacian_2d(u, DX))
    out["visc_u"] = m(viscosity.incompressible_viscosity_2d(u, C.mu_air, DX))
    out["mu_air"] = m(C.mu_air)
    for n in range(1, 6):
        u, v, p, t = matsumo_temp.matsumo_scheme(u, v, p, t, DX, DT)
        if n in (1, 5):
            out["u%d" % n], out["v%d" % n], out["p%d" % n], out["t%d" % n] = m(u), m(v), m(p), m(t)
    save("g3_sw2d_temp", **out)


# --------------------------------------------------------------------- G4
def g4_tracer():
    rng = np.random.default_rng(4)
    shape = (8, 12)
    V0 = rng.standard_normal((2,) + shape) * 2
    q0 = rng.random(shape)
    p0 = 101325 + rng.standard_normal(shape)
    t0 = 273.15 + rng.standard_normal(shape)
    dt, sc = 1.0, (10.0, 12.5)
    V, q, p, t = V0 * MS, q0 * U.kg, p0 * U.Pa, t0 * U.K
    DT, SC = dt * U.s, (sc[0] * U.m, sc[1] * U.m)
    out = dict(V0=V0, q0=q0, p0=p0, t0=t0, dt=dt, sc=np.asarray(sc))
    for ax in (0, 1):
        out["upwind_axis%d" % ax] = m(two_d.upwind_axis(DT, SC, V, q, ax))
        out["upwind_axis_finite%d" % ax] = m(two_d.upwind_axis_finite(DT, SC, V, q, ax))
        out["fv_upwind%d" % ax] = m(two_d.fv_advect_axis_upwind(DT, SC, V, q, ax))
        out["fv_upwind_finite%d" % ax] = m(two_d.fv_advect_axis_upwind_finite(DT, SC, V, q, ax))
        out["fv_plain%d" % ax] = m(two_d.fv_advect_axis_plain(DT, SC, V, q, ax))
        out["fv_plain_finite%d" % ax] = m(two_d.fv_advect_axis_plain_finite(DT, SC, V, q, ax))
        out["pgf_axis%d" % ax] = m(two_d.pgf_c_grid_axis(p, SC, ax))
    out["ctu"] = m(two_d.corner_transport_2d(DT, SC, V, q))
    out["fva"] = m(two_d.finite_volume_advection(DT, SC, V, q))
    out["pgf_c_grid"] = m(two_d.pgf_c_grid(DT, SC, p, t))
    out["pgf_templess"] = m(two_d.pgf_templess(DT, SC, p))
    out["p_edge"] = m(two_d.pressure_at_edge(p))
    out["p_edge_1d"] = m(two_d.pressure_at_edge_one_d(p))
    out["adv_mom"] = m(two_d.advect_with_momentum(DT, SC, V, p))
    out["pgf_one_d"] = m(two_d.pgf_one_d(DT, SC[0], p))
    out["pgf_one_d_axis1"] = m(two_d.pgf_one_d(DT, SC[1], p, 1))
    out["pgf_one_d_line"] = m(two_d.pgf_one_d(DT, SC[0], p[:, 0]))
    for ax in (0, 1):
        out["gradient%d" % ax] = m(two_d.gradient(p, SC, ax))
    out["pressure_gradient"] = m(two_d.pressure_gradient(DT, SC, p, t))
    # 1-D limiter pieces
    q1 = rng.random(16)
    q1[3] = q1[4] = q1[5]          # exact-zero denominators for calc_r
    u1 = rng.standard_normal(16) * 5
    u1[7] = 0.0                    # strict `u > 0`
    r = flux_limiter.calc_r(q1 * U.kg)
    out.update(q1=q1, u1=u1, r1=m(r), phi1=m(flux_limiter.van_leer(r)),
               phi_pts=np.asarray([flux_limiter.van_leer(x) for x in (1, 0, -2.0, 0.5, 1e30)]),
               donor_flux=m(flux_limiter.donor_cell_flux(q1 * U.kg, u1 * MS)),
               donor_adv=m(flux_limiter.donor_cell_advection(q1 * U.kg, u1 * MS, 100 * U.m, 1 * U.s)))
    # the test_2d.py:240-252 state-dict setup, 400 steps of ft
    side, half, quarter = 4, 2, 1
    Vt = np.zeros((2, side, side)) * MS
    qt = np.zeros((side, side)) * U.kg
    qt[quarter:half, quarter:half] = 1.0 * U.kg
    Vt[0][:] = 2.0 * MS
    Vt[1][:] = -2.0 * MS
    tv = [m(C.get_total_variation(qt))]
    qq = qt
    for _ in range(400):
        qq = two_d.corner_transport_2d(1 * U.s, (10 * U.m, 10 * U.m), Vt, qq)
        tv.append(m(C.get_total_variation(qq)))
    out["runfunc_q400"] = m(qq)
    out["runfunc_tv"] = np.asarray(tv)
    qq = qt
    for _ in range(400):
        qq = two_d.finite_volume_advection(1 * U.s, (10 * U.m, 10 * U.m), Vt, qq)
    out["fv_q400"] = m(qq)
    save("g4_tracer", **out)


# --------------------------------------------------------------------- G5
GEOM_KEYS = ("sige", "sigt", "sigb", "dsig", "sig", "dsigv", "dx_j", "dx_h", "dy",
             "ptop", "heightmap")


def geom_arrays(geom, prefix):
    out = {}
    for k in GEOM_KEYS + ("area", "lat", "long"):
        if hasattr(geom, k):
            out[prefix + k] = m(getattr(geom, k))
    return out


def g5_geometry():
    out = {}
    for (h, w, l) in ((24, 36, 9), (8, 16, 4), (12, 20, 5)):
        for sname in ("manabe_sig", "equal_sig"):
            g = quiet(geometry.gen_geometry, h, w, l, sig_func=getattr(geometry, sname))
            out.update(geom_arrays(g, "g_%d_%d_%d_%s_" % (h, w, l, sname)))
    g = quiet(geometry.gen_geometry, 720, 1440, 24, sig_func=geometry.manabe_sig)
    big = geom_arrays(g, "g_720_1440_24_manabe_sig_")
    big.pop("g_720_1440_24_manabe_sig_heightmap")
    out.update(big)
    g = quiet(geometry.gen_square_geometry, 6, 10, 3, 1000.0 * U.m, 1200.0 * U.m)
    out.update(geom_arrays(g, "sq_6_10_3_"))
    save("g5_geometry", **out)


# --------------------------------------------------------------------- G6
def g6_lowpass():
    rng = np.random.default_rng(6)
    out = {}
    for (l, h, w) in ((3, 8, 16), (9, 24, 36), (2, 6, 10)):
        g = quiet(geometry.gen_geometry, h, w, l)
        a = rng.standard_normal((l, h, w))
        out["in_%d_%d_%d" % (l, h, w)] = a
        out["out_%d_%d_%d" % (l, h, w)] = m(low_pass.arakawa_1977(a * U.m, g))
    save("g6_lowpass", **out)


# --------------------------------------------------------------------- G7/G8
def dense_ic(geom, rng, wind=1.0):
    L, H, W = geom.layers, geom.height, geom.width
    p = 1e5 + 10 * rng.standard_normal((H, W))
    u = wind * rng.standard_normal((L, H, W))
    v = wind * rng.standard_normal((L, H, W))
    v[:, -1, :] = 0
    tt = 300 + rng.standard_normal((L, H, W))
    tp = p * m(geom.sig) + m(geom.ptop)
    t = m(temperature.to_potential_temp(tt * U.K, tp * U.Pa))
    q = 3e-6 * (1 + 0.1 * rng.random((L, H, W)))
    return p, u, v, t, q


def g7_half_step():
    rng = np.random.default_rng(7)
    L, H, W = 5, 12, 20
    geom = quiet(geometry.gen_geometry, H, W, L, sig_func=geometry.manabe_sig)
    geom.heightmap[3, 4] = 500 * U.m          # exercise the heightmap*G term
    p0, u0, v0, t0, q0 = dense_ic(geom, rng)
    dt = 300.0
    out = dict(p0=p0, u0=u0, v0=v0, t0=t0, q0=q0, dt=dt, heightmap=m(geom.heightmap))
    base = (p0 * U.Pa, u0 * MS, v0 * MS, t0 * U.K, q0 * U.dimensionless)
    stage = base
    for tag in ("pred", "corr"):
        p, u, v, t, q = base
        sp, su, sv, st, sq = stage
        spu_orig = dynamics.calc_pu(sp, su)
        spu = low_pass.arakawa_1977(spu_orig, geom)
        spv = dynamics.calc_pv(sp, sv)
        pit, sd = dynamics.aflux(spu, spv, geom)
        dut, dvt = dynamics.advec_m_pu(sp, su, sv, spu, spv, geom)
        pgu, pgv, phiu, phiv = quiet(dynamics.pgf, sp, st, geom)
        phi = quiet(dynamics.compute_geopotential, sp, st, geom)
        dus = dynamics.advec_sig(c3.iph(sd), su, geom)
        dvs = dynamics.advec_sig(c3.jph(sd), sv, geom)
        pgfu = low_pass.arakawa_1977(pgu + phiu, geom)
        adt = dynamics.advec_t(spu, spv, st, geom)
        ads = dynamics.advec_sig(sd, st, geom)
        nxt = quiet(dynamics.half_timestep, p, u, v, t, q, sp, su, sv, st, sq, dt * U.s, geom)
        for k, val in dict(spu_orig=spu_orig, spu=spu, spv=spv, pit=pit, sd=sd, dut=dut,
                           dvt=dvt, pgu=pgu, pgv=pgv, phiu=phiu, phiv=phiv, phi=phi,
                           dus=dus, dvs=dvs, pgfu=pgfu, advec_t=adt, advec_sig_t=ads,
                           p_n=nxt[0], u_n=nxt[1], v_n=nxt[2], t_n=nxt[3], q_n=nxt[4]).items():
            out["%s_%s" % (tag, k)] = m(val)
        stage = nxt
    full = quiet(dynamics.matsuno_timestep, *base, dt * U.s, geom)
    for k, val in zip("puvtq", full):
        out["step_" + k] = m(val)
    save("g7_half_step", **out)


def g8_pe25d():
    out = {}
    # (a) the reference harness run_model (no_limits_2_5d.py:220-236) at its own main()
    # size 8x8x3, dt = 1800 s (:260).  NB calc_energy broadcasts geom.area (H,) against
    # (L,H,W) (:49), so run_model/full_timestep only run when W == H (or H == 1).
    no_limits_2_5d.STATS.clear()
    snaps = {}

    def cb(p, u, v, t, q, _n=[0]):
        _n[0] += 1
        if _n[0] in (1, 3, 10):
            snaps[_n[0]] = [m(x).copy() for x in (p, u, v, t, q)]

    quiet(no_limits_2_5d.run_model, 8, 8, 3, 1800 * U.s, 10, cb)
    for n, st in snaps.items():
        for k, val in zip("puvtq", st):
            out["harness%d_%s" % (n, k)] = val
    S = no_limits_2_5d.STATS
    for k in ("u_max", "u_min", "v_max", "v_min"):
        out["harness_stats_" + k] = np.asarray([float(m(x)) for x in S[k]])
    out["harness_stats_ke"] = np.asarray([[float(m(y)) for y in x] for x in S["ke"]])
    # (a2) the same initial condition at 36x24x9, dt = 900 s, through matsuno_timestep
    geom = quiet(geometry.gen_geometry, 24, 36, 9, sig_func=geometry.manabe_sig)
    p, u, v, t, q, gr = no_limits_2_5d.gen_initial_conditions(geom)
    v[0, 0, 0] = 0.1 * v.u
    u *= 0
    st = (p, u, v, t, q)
    for n in range(1, 11):
        st = quiet(dynamics.matsuno_timestep, *st, 900.0 * U.s, geom)
        if n in (1, 3, 10):
            for k, val in zip("puvtq", st):
                out["h36_%d_%s" % (n, k)] = m(val)
    # (b) gen_initial_conditions
    for (h, w, l) in ((24, 36, 9), (8, 16, 4)):
        g = quiet(geometry.gen_geometry, h, w, l, sig_func=geometry.manabe_sig)
        p, u, v, t, q, gr = no_limits_2_5d.gen_initial_conditions(g)
        for k, val in zip(("p", "u", "v", "t", "q", "gt"), (p, u, v, t, q, gr.gt)):
            out["ic_%d_%d_%d_%s" % (h, w, l, k)] = m(val)
    # (c) dense random IC, 1/3/10 steps, 36x24x9, dt = 300 s
    rng = np.random.default_rng(8)
    geom = quiet(geometry.gen_geometry, 24, 36, 9, sig_func=geometry.manabe_sig)
    p0, u0, v0, t0, q0 = dense_ic(geom, rng)
    out.update(dense_p0=p0, dense_u0=u0, dense_v0=v0, dense_t0=t0, dense_q0=q0, dense_dt=300.0)
    st = (p0 * U.Pa, u0 * MS, v0 * MS, t0 * U.K, q0 * U.dimensionless)
    for n in range(1, 11):
        st = quiet(dynamics.matsuno_timestep, *st, 300.0 * U.s, geom)
        if n in (1, 3, 10):
            for k, val in zip("puvtq", st):
                out["dense%d_%s" % (n, k)] = m(val)
    assert all(np.isfinite(m(x)).all() for x in st), "dense IC went non-finite"
    # (d) the test_geography.py:6-23,49 harness: H=1, W=16, L=17, 1000 m bump at [0, 8]
    geom = quiet(geometry.gen_geometry, 1, 16, 17, sig_func=geometry.manabe_sig)
    p, u, v, t, q, gr = no_limits_2_5d.gen_initial_conditions(geom)
    v[0, 0, 0] = 0.1 * v.u
    u *= 0
    geom.heightmap.m[0, 8] = 1000
    st = (p, u, v, t, q)
    for n in range(3):
        st = quiet(dynamics.matsuno_timestep, *st, 1800.0 * U.s, geom)
    for k, val in zip("puvtq", st):
        out["bump3_" + k] = m(val)
    e = no_limits_2_5d.calc_energy(*st, gr, geom)
    out["bump3_energy"] = np.asarray([float(m(x)) for x in e])
    save("g8_pe25d", **out)


# --------------------------------------------------------------------- G9 + PE2D
def g9_oned():
    out = {}
    side = 128
    p = np.full(side, 1) * C.standard_pressure
    u = np.full(side, 1) * 1.0 * MS
    q = np.full(side, 1) * 0.1 * U.dimensionless
    t = np.full(side, 1) * temperature.to_potential_temp(C.standard_temperature, p)
    p[3] *= 1.00001
    q[side // 4:side // 2] = 1
    out.update(p0=m(p).copy(), u0=m(u).copy(), t0=m(t).copy(), q0=m(q).copy(), dx=70000.0, dt=100.0)
    # the reference's own dt = 900 s (no_limits.py:215) blows up to NaN by step 5 with
    # this IC; 2 steps at 900 s and 10 steps at 100 s are kept as vectors
    st = (p, u, t, q)
    for _ in range(2):
        st = no_limits.matsuno_timestep(*st, 900.0 * U.s, 70000 * U.m)
    out.update(p2_900=m(st[0]), u2_900=m(st[1]), t2_900=m(st[2]), q2_900=m(st[3]))
    for _ in range(10):
        p, u, t, q = no_limits.matsuno_timestep(p, u, t, q, 100.0 * U.s, 70000 * U.m)
    out.update(p10=m(p), u10=m(u), t10=m(t), q10=m(q))
    # the operators of the step one by one (no_limits.py:50-112), on the state after those 10 steps
    DX1 = 70000 * U.m
    pu1 = no_limits.calc_pu(u, p)
    out.update(op_p=m(p), op_u=m(u), op_t=m(t), op_q=m(q), op_calc_pu=m(pu1), op_un_pu=m(no_limits.un_pu(pu1, p)),
               op_advec_q=m(no_limits.advec_q(u, q, DX1)), op_advec_p=m(no_limits.advec_p(pu1, DX1)),
               op_advec_pu=m(no_limits.advec_pu(p, pu1, u, DX1)), op_advec_t=m(no_limits.advec_t(pu1, t, DX1)),
               op_pgf=m(no_limits.pgf(p, t, DX1)))
    # 161-cell upwind mass advection, 400 steps (test_oneD.py world_shape)
    n = 161
    V = np.full((1, n), 2.0) * MS
    qq = np.zeros(n) * U.kg
    qq[n // 4:n // 2] = 1.0 * U.kg
    out["adv_q0"] = m(qq).copy()
    for _ in range(400):
        qq = two_d.fv_advect_axis_upwind(1 * U.s, (10 * U.m,), V, qq, 0)
    out["adv_q400"] = m(qq)
    save("g9_oned", **out)


def g10_pe2d():
    rng = np.random.default_rng(10)
    shape = (16, 32)
    p0 = 101325 + 10 * rng.standard_normal(shape)
    u0 = rng.standard_normal(shape)
    v0 = rng.standard_normal(shape)
    t0 = 300 + rng.standard_normal(shape)
    q0 = rng.random(shape)
    dx, dt = 100e3, 100.0
    p, u, v, t, q = p0 * U.Pa, u0 * MS, v0 * MS, t0 * U.K, q0 * U.dimensionless
    DX, DT = dx * U.m, dt * U.s
    out = dict(p0=p0, u0=u0, v0=v0, t0=t0, q0=q0, dx=dx, dt=dt)
    pu, pv = no_limits_2d.calc_pu(p, u), no_limits_2d.calc_pv(p, v)
    out["pu"], out["pv"] = m(pu), m(pv)
    out["advec_p"] = m(no_limits_2d.advec_p(pu, pv, DX))
    out["dut"], out["dvt"] = m(no_limits_2d.advec_m(p, u, v, DX))
    out["pgu"], out["pgv"] = m(no_limits_2d.pgf(p, t, DX))
    out["advec_t"] = m(no_limits_2d.advec_t(pu, pv, t, DX))
    st = (p, u, v, t, q)
    for n in range(1, 6):
        st = no_limits_2d.matsuno_timestep(*st, DT, DX)
        if n in (1, 5):
            for k, val in zip("puvtq", st):
                out["%s%d" % (k, n)] = m(val)
    save("g10_pe2d", **out)


def g11_temperature():
    rng = np.random.default_rng(11)
    tt = 250 + 50 * rng.random((4, 6))
    p = 2e4 + 8e4 * rng.random((4, 6))
    th = temperature.to_potential_temp(tt * U.K, p * U.Pa)
    save("g11_temperature", tt=tt, p=p, theta=m(th),
         back=m(temperature.to_true_temp(th, p * U.Pa)),
         rho=m(temperature.to_density(tt * U.K, p * U.Pa)),
         kappa=m(C.kappa), P0=m(C.P0), G=m(C.G), Rd=m(C.Rd), Cp=m(C.Cp),
         radius=m(C.radius), Rv=m(C.Rv),
         std_roundtrip=np.asarray([float(m(temperature.to_true_temp(
             temperature.to_potential_temp(C.standard_temperature, C.standard_pressure),
             C.standard_pressure)))]))


def g12_coriolis():
    """The reference keeps its Coriolis terms behind a literal `if False:` (dynamics.py:82).
    To pin the build's optional Coriolis path against the author's own expressions, the module
    source is read as text, that one literal is flipped to `if True:` IN MEMORY and the result is
    executed as a module (nothing is written to disk); everything else is the unmodified file."""
    import types
    src = open(os.path.join(REF, "dynamics.py")).read()
    assert src.count("    if False:\n        pu_at_pv = imh(jph(pu))") == 1
    src = src.replace("    if False:\n        pu_at_pv = imh(jph(pu))", "    if True:\n        pu_at_pv = imh(jph(pu))")
    mod = types.ModuleType("dynamics_coriolis_on")
    mod.__file__ = os.path.join(REF, "dynamics.py")
    with contextlib.redirect_stdout(_sink):
        exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    rng = np.random.default_rng(12)
    L, H, W = 5, 12, 20
    geom = quiet(geometry.gen_geometry, H, W, L, sig_func=geometry.manabe_sig)
    p0, u0, v0, t0, q0 = dense_ic(geom, rng)
    st = (p0 * U.Pa, u0 * MS, v0 * MS, t0 * U.K, q0 * U.dimensionless)
    out = dict(p0=p0, u0=u0, v0=v0, t0=t0, q0=q0, dt=300.0)
    dut, dvt = mod.advec_m_pu(st[0], st[1], st[2], dynamics.calc_pu(st[0], st[1]),
                              dynamics.calc_pv(st[0], st[2]), geom)
    out["dut"], out["dvt"] = m(dut), m(dvt)
    for n in (1, 2):
        st = quiet(mod.matsuno_timestep, *st, 300.0 * U.s, geom)
        for k, val in zip("puvtq", st):
            out["step%d_%s" % (n, k)] = m(val)
    save("g12_coriolis", **out)


def g13_radiation():
    """grey_solar.basic_grey_radiation and no_limits_2_5d.solar_timestep on a dense state"""
    import grey_solar
    rng = np.random.default_rng(13)
    L, H, W = 5, 12, 20
    geom = quiet(geometry.gen_geometry, H, W, L, sig_func=geometry.manabe_sig)
    p0, u0, v0, t0, q0 = dense_ic(geom, rng)
    gt0 = 290 + 5 * rng.standard_normal((H, W))
    out = dict(p0=p0, t0=t0, gt0=gt0, dt=900.0)
    g = no_limits_2_5d.GroundVars(gt0 * U.K, None, None, None)
    for tag, utc in (("a", 0.0), ("b", 7.5 * 3600.0)):
        tp = p0 * U.Pa * geom.sig + geom.ptop
        tt = temperature.to_true_temp(t0 * U.K, tp)
        dTdt, dtg = quiet(grey_solar.basic_grey_radiation, p0 * U.Pa, tp, tt, g, 0.1, 0.9, 0.3, utc * U.s, geom)
        t_n, g_n = quiet(no_limits_2_5d.solar_timestep, t0 * U.K, p0 * U.Pa, g, 900.0 * U.s, utc * U.s, geom)
        out.update({"utc_" + tag: utc, "dTdt_" + tag: m(dTdt), "dtg_" + tag: m(dtg), "t_n_" + tag: m(t_n),
                    "gt_n_" + tag: m(g_n.gt), "sza_" + tag: m(grey_solar.zenith_angle(geom.long, geom.lat, utc * U.s, geom))})
    save("g13_radiation", **out)


def g14_humidity():
    """humidity.py: manabe_rh, the Buck saturation vapour pressure, mixing ratios and the rh <-> mmr pair"""
    with contextlib.redirect_stdout(_sink):
        import humidity
    rng = np.random.default_rng(14)
    geom = quiet(geometry.gen_geometry, 6, 8, 9, sig_func=geometry.manabe_sig)
    n = 4000
    tt = 200.0 + 130.0 * rng.random(n)                    # K
    tp = 2000.0 + 101000.0 * rng.random(n)                # Pa
    rh = 0.01 + 0.99 * rng.random(n)
    # keep the vapour pressure below the total pressure (hot air at low pressure has no such state)
    es = m(humidity.saturation_vapor_pressure(tt * U.K))
    keep = es < 0.5 * tp
    tt, tp, rh = tt[keep], tp[keep], rh[keep]
    svp = humidity.saturation_vapor_pressure(tt * U.K)
    ws = humidity.w_s_at(tp * U.Pa, tt * U.K)
    mmr = humidity.rh_to_mmr(rh, tp * U.Pa, tt * U.K)
    back = humidity.mmr_to_rh(mmr, tp * U.Pa, tt * U.K)
    vmr = humidity.vmr_from_mmr(mmr, C.M_water, C.Md)
    save("g14_humidity", sig=m(geom.sig), manabe_rh=m(humidity.manabe_rh(geom)), tt=tt, tp=tp, rh=rh,
         svp=m(svp), w_s=m(ws), mmr=m(mmr), rh_back=m(back), vmr=m(vmr),
         M_water=m(C.M_water), Md=m(C.Md))


if __name__ == "__main__":
    for f in (g1_shifts, g2_sw2d, g3_sw2d_temp, g4_tracer, g5_geometry, g6_lowpass,
              g7_half_step, g8_pe25d, g9_oned, g10_pe2d, g11_temperature, g12_coriolis, g13_radiation, g14_humidity):
        if len(sys.argv) > 1 and f.__name__ not in sys.argv[1:]:
            continue                                      # python make_golden.py g14_humidity: only that set
        f()
