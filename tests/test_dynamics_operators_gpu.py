"""GPU parity of the dynamics.py operators one by one (gcm_pe25d_op) vs the golden vectors G7: every
intermediate of the reference's half_timestep (predictor and corrector), captured from the reference
run on 20x12x5 with a topography bump -- and vs the oracle on other shapes."""
import numpy as np
import pytest

from conftest import golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _iph(x): return (x + np.roll(x, -1, -1)) / 2
def _jph(x): return (x + np.roll(x, -1, -2)) / 2


@pytest.mark.parametrize("stage", ["pred", "corr"])
def test_every_intermediate_of_a_half_step_vs_golden(stage):
    from gcmiipy_amd import geometry, dynamics as dyn
    d = golden("g7_half_step")
    L, H, W = d["u0"].shape
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    geom.heightmap[...] = d["heightmap"]
    base = [d[k + "0"] for k in "puvtq"]
    sp, su, sv, st, sq = base if stage == "pred" else [d["pred_%s_n" % k] for k in "puvtq"]
    g = lambda k: d["%s_%s" % (stage, k)]
    assert rel_err(dyn.calc_pu(sp, su), g("spu_orig")) < TOL
    assert rel_err(dyn.calc_pv(sp, sv), g("spv")) < TOL
    spu, spv = g("spu"), g("spv")                                   # (spu: filtered, test_low_pass_gpu.py)
    pit, sd = dyn.aflux(spu, spv, geom)
    assert rel_err(pit, g("pit")) < TOL and rel_err(sd, g("sd")) < TOL
    dut, dvt = dyn.advec_m_pu(sp, su, sv, spu, spv, geom)
    assert rel_err(dut, g("dut")) < TOL and rel_err(dvt, g("dvt")) < TOL
    assert rel_err(dyn.compute_geopotential(sp, st, geom), g("phi")) < TOL
    for got, key in zip(dyn.pgf(sp, st, geom), ("pgu", "pgv", "phiu", "phiv")):
        assert rel_err(got, g(key)) < TOL, key
    sdg = g("sd")
    assert rel_err(dyn.advec_sig(_iph(sdg), su, geom), g("dus")) < TOL
    assert rel_err(dyn.advec_sig(_jph(sdg), sv, geom), g("dvs")) < TOL
    assert rel_err(dyn.advec_t(spu, spv, st, geom), g("advec_t")) < TOL
    assert rel_err(dyn.advec_sig(sdg, st, geom), g("advec_sig_t")) < TOL
    # un_pu / un_pv undo calc_pu / calc_pv on the new pressure: u_n = un_pu(pu_n, p_n) (dynamics.py:216-217)
    p_n, u_n, v_n = g("p_n"), g("u_n"), g("v_n")
    assert rel_err(dyn.un_pu(dyn.calc_pu(p_n, u_n), p_n), u_n) < TOL
    assert rel_err(dyn.un_pv(dyn.calc_pv(p_n, v_n), p_n), v_n) < TOL


@pytest.mark.parametrize("hwl", [(6, 8, 1), (5, 130, 3), (24, 36, 9)])
def test_operators_vs_oracle_shapes(hwl):
    from gcmiipy_amd import geometry, dynamics as dyn
    from oracle import dynamics as od, geometry as ogeo, temperature as otemp
    H, W, L = hwl
    rng = np.random.default_rng(H * W + L)
    geom = geometry.gen_geometry(H, W, L, sig_func=geometry.manabe_sig)
    og = ogeo.gen_geometry(H, W, L, sig_func=ogeo.manabe_sig)
    geom.heightmap[...] = og.heightmap[...] = 50 * rng.random((H, W))
    p = 1e5 + 100 * rng.standard_normal((H, W))
    u, v = rng.standard_normal((L, H, W)), rng.standard_normal((L, H, W))
    t = otemp.to_potential_temp(300 + rng.standard_normal((L, H, W)), p * og.sig + og.ptop)
    pu, pv = od.calc_pu(p, u), od.calc_pv(p, v)
    assert rel_err(dyn.calc_pu(p, u), pu) < TOL and rel_err(dyn.calc_pv(p, v), pv) < TOL
    for got, want in zip(dyn.aflux(pu, pv, geom), od.aflux(pu, pv, og)):
        assert rel_err(got, want) < TOL
    sd = od.aflux(pu, pv, og)[1]
    assert rel_err(dyn.advec_sig(sd, t, geom), od.advec_sig(sd, t, og)) < TOL
    for got, want in zip(dyn.advec_m_pu(p, u, v, pu, pv, geom), od.advec_m_pu(p, u, v, pu, pv, og)):
        assert rel_err(got, want) < TOL
    assert rel_err(dyn.compute_geopotential(p, t, geom), od.compute_geopotential(p, t, og)) < TOL
    for got, want in zip(dyn.pgf(p, t, geom), od.pgf(p, t, og)):
        assert rel_err(got, want) < 1e-9                            # (gradients of phi ~ 1e5: differences of large numbers)
    assert rel_err(dyn.advec_t(pu, pv, t, geom), od.advec_t(pu, pv, t, og)) < TOL
    with pytest.raises(ValueError):
        dyn.aflux(pu[:, :, :-1], pv, geom)
