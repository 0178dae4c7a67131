"""Host-side geometry tables of the drop-in must equal the reference's BIT FOR BIT
(these are the "geometry" items the kernels read)."""
import numpy as np
import pytest

from conftest import golden
from gcmiipy_amd import geometry

KEYS = ("sige", "sigt", "sigb", "dsig", "sig", "dsigv", "dx_j", "dx_h", "dy", "ptop",
        "heightmap", "area", "lat", "long")


@pytest.mark.parametrize("hwl", [(24, 36, 9), (8, 16, 4), (12, 20, 5)])
@pytest.mark.parametrize("sname", ["manabe_sig", "equal_sig"])
def test_gen_geometry_bit_exact(hwl, sname):
    d = golden("g5_geometry")
    h, w, l = hwl
    g = geometry.gen_geometry(h, w, l, sig_func=getattr(geometry, sname))
    for k in KEYS:
        a, b = np.asarray(getattr(g, k)), d["g_%d_%d_%d_%s_%s" % (h, w, l, sname, k)]
        assert a.shape == b.shape and np.array_equal(a, b), k


def test_large_and_square_geometry_bit_exact():
    d = golden("g5_geometry")
    g = geometry.gen_geometry(720, 1440, 24, sig_func=geometry.manabe_sig)
    for k in KEYS:
        if k != "heightmap":
            assert np.array_equal(np.asarray(getattr(g, k)), d["g_720_1440_24_manabe_sig_" + k]), k
    g = geometry.gen_square_geometry(6, 10, 3, 1000.0, 1200.0)
    for k in ("sige", "sigt", "sigb", "dsig", "sig", "dsigv", "dx_j", "dx_h", "dy", "ptop", "heightmap"):
        assert np.array_equal(np.asarray(getattr(g, k)), d["sq_6_10_3_" + k]), k


def test_initial_conditions_bit_exact():
    """no_limits_2_5d.gen_initial_conditions (host NumPy) vs arrays captured from the reference"""
    from gcmiipy_amd import no_limits_2_5d as h
    d = golden("g8_pe25d")
    for (hh, w, l) in ((24, 36, 9), (8, 16, 4)):
        g = geometry.gen_geometry(hh, w, l, sig_func=geometry.manabe_sig)
        p, u, v, t, q, gr = h.gen_initial_conditions(g)
        for k, x in zip(("p", "u", "v", "t", "q", "gt"), (p, u, v, t, q, gr.gt)):
            assert np.array_equal(x, d["ic_%d_%d_%d_%s" % (hh, w, l, k)]), k
