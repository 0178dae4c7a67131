"""GPU parity of the two_d.py operator drop-ins vs the golden vectors G4 (reference
two_d.py run on seeded inputs) and the reference's own TV tests (test_2d.py:47-80,140-173)."""
import numpy as np
import pytest

from conftest import golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_two_d_operators_vs_golden():
    from gcmiipy_amd import two_d
    d = golden("g4_tracer")
    V, q, p, t, dt = d["V0"], d["q0"], d["p0"], d["t0"], float(d["dt"])
    sc = tuple(float(x) for x in d["sc"])
    for ax in (0, 1):
        for name, fn in (("upwind_axis", two_d.upwind_axis), ("upwind_axis_finite", two_d.upwind_axis_finite),
                         ("fv_upwind", two_d.fv_advect_axis_upwind),
                         ("fv_upwind_finite", two_d.fv_advect_axis_upwind_finite),
                         ("fv_plain", two_d.fv_advect_axis_plain),
                         ("fv_plain_finite", two_d.fv_advect_axis_plain_finite)):
            got = fn(dt, sc, V, q, ax)
            want = d["%s%d" % (name, ax)]
            scale = max(np.max(np.abs(want)), np.max(np.abs(q)) * 1e-3)
            assert np.max(np.abs(got - want)) / scale < TOL, (name, ax)
        assert rel_err(two_d.pgf_c_grid_axis(p, sc, ax), d["pgf_axis%d" % ax]) < TOL
    assert rel_err(two_d.corner_transport_2d(dt, sc, V, q), d["ctu"]) < TOL
    assert rel_err(two_d.finite_volume_advection(dt, sc, V, q), d["fva"]) < TOL
    assert rel_err(two_d.pgf_c_grid(dt, sc, p, t), d["pgf_c_grid"]) < TOL
    assert rel_err(two_d.pgf_templess(dt, sc, p), d["pgf_templess"]) < TOL
    assert rel_err(two_d.pressure_at_edge(p), d["p_edge"]) < TOL
    assert rel_err(two_d.pressure_at_edge_one_d(p), d["p_edge_1d"]) < TOL
    assert rel_err(two_d.pressure_at_edge_one_d(p[:, 3]), d["p_edge_1d"][:, 3]) < TOL      # a 1-D line
    assert rel_err(two_d.pgf_one_d(dt, sc[0], p), d["pgf_one_d"]) < TOL
    assert rel_err(two_d.pgf_one_d(dt, sc[1], p, 1), d["pgf_one_d_axis1"]) < TOL
    assert rel_err(two_d.pgf_one_d(dt, sc[0], p[:, 0]), d["pgf_one_d_line"]) < TOL
    for ax in (0, 1):
        assert rel_err(two_d.gradient(p, sc, ax), d["gradient%d" % ax]) < TOL
    assert rel_err(two_d.pressure_gradient(dt, sc, p, t), d["pressure_gradient"]) < TOL
    with pytest.raises(ValueError):
        two_d.pgf_one_d(dt, sc[0], p[:, 0], 1)
    with pytest.raises(ValueError):
        two_d.pressure_gradient(dt, sc, p, None)
    assert rel_err(two_d.advect_with_momentum(dt, sc, V, p), d["adv_mom"]) < TOL


def test_reference_kats_on_gpu():
    from gcmiipy_amd import two_d
    p = np.full((3, 3), 1.0)
    p[1, 1] = 0
    assert two_d.pressure_at_edge(p)[0][0][1] == 0.5              # test_2d.py:176-181


def test_state_dict_driver_and_tv():
    """test_2d.py:240-252 (test_run_func) through run_2d_with_ft; TV never grows (:69-72)."""
    from gcmiipy_amd import two_d
    d = golden("g4_tracer")
    V = np.zeros((2, 4, 4)); q = np.zeros((4, 4))
    q[1:2, 1:2] = 1.0
    V[0][:] = 2.0; V[1][:] = -2.0

    def ft(V, q):
        return {"V": V, "q": two_d.corner_transport_2d(1.0, (10.0, 10.0), V, q)}

    hist = []
    assert two_d.run_2d_with_ft({"V": V, "q": q}, ft, history=hist) is True
    assert rel_err(two_d.run_2d_with_ft.last_state["q"], d["runfunc_q400"]) < TOL
    assert max(hist) <= hist[0] + 0.00001
    assert np.max(np.abs(np.asarray(hist) - d["runfunc_tv"])) < 1e-9
    # 400 device-resident steps of finite_volume_advection (test_2d.py:140-173)
    got = two_d.finite_volume_advection(1.0, (10.0, 10.0), V, q, steps=400)
    assert rel_err(got, d["fv_q400"]) < TOL


def test_limited_advection_vs_oracle_and_1d():
    from gcmiipy_amd import two_d
    from oracle import tracer
    d = golden("g4_tracer")
    V, q, dt = d["V0"], d["q0"], float(d["dt"])
    sc = tuple(float(x) for x in d["sc"])
    want = q
    for _ in range(5):
        want = tracer.limited_advection(dt, sc, V, want)
    assert rel_err(two_d.limited_advection(dt, sc, V, q, steps=5), want) < TOL
    g9 = golden("g9_oned")                                         # 161-cell 1-D upwind, 400 steps
    qq = g9["adv_q0"]
    Vv = np.full((1, 161), 2.0)
    for _ in range(3):
        qq = two_d.fv_advect_axis_upwind(1.0, (10.0,), Vv, qq, 0)
    ref = g9["adv_q0"]
    for _ in range(3):
        ref = tracer.fv_advect_axis_upwind(1.0, (10.0,), Vv, ref, 0)
    assert rel_err(qq, ref) < TOL
