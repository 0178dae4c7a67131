"""GPU parity of the polar-filter drop-in gcmiipy_amd.low_pass.arakawa_1977 (reference
low_pass.py:41-78) vs the golden vectors G6 (the reference's own rfft / irfft) and vs the oracle at
the row lengths of the BASELINE grids."""
import numpy as np
import pytest

from conftest import golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.mark.parametrize("lhw", [(3, 8, 16), (9, 24, 36), (2, 6, 10)])
def test_arakawa_1977_vs_golden(lhw):
    from gcmiipy_amd import geometry, low_pass
    d = golden("g6_lowpass")
    l, h, w = lhw
    geom = geometry.gen_geometry(h, w, l)
    q = d["in_%d_%d_%d" % lhw]
    want = d["out_%d_%d_%d" % lhw]
    assert rel_err(low_pass.arakawa_1977(q, geom), want) < TOL
    assert rel_err(low_pass.arakawa_1977(q[0], geom), want[0]) < TOL          # a 2-D field
    # more levels than the handle has layers: filtered `layers` at a time
    q5 = np.concatenate([q, q[::-1]])
    assert rel_err(low_pass.arakawa_1977(q5, geom), np.concatenate([want, want[::-1]])) < TOL


@pytest.mark.parametrize("hwl", [(12, 1440, 3), (8, 2880, 2), (6, 4096, 1), (10, 14, 2)])
def test_arakawa_1977_vs_oracle_baseline_row_lengths(hwl):
    """1440 / 2880 / 4096: the in-place composite-radix plans; 14 = 2.7: the generic path"""
    from gcmiipy_amd import geometry, low_pass
    from oracle import lowpass, geometry as ogeo
    h, w, l = hwl
    rng = np.random.default_rng(w)
    q = rng.standard_normal((l, h, w)) * 1e3
    got = low_pass.arakawa_1977(q, geometry.gen_geometry(h, w, l))
    assert rel_err(got, lowpass.arakawa_1977(q, ogeo.gen_geometry(h, w, l))) < TOL
    # the zonal mean passes unchanged, and filtering twice damps further or leaves alone: |S| <= 1
    assert np.allclose(got.mean(axis=-1), q.mean(axis=-1), rtol=0, atol=1e-9)
    assert np.linalg.norm(got) <= np.linalg.norm(q) * (1 + 1e-12)


def test_multiplier_table_and_errors():
    from gcmiipy_amd import geometry, low_pass
    from oracle import lowpass, geometry as ogeo
    for h, w in ((8, 16), (24, 36), (6, 10)):
        assert np.array_equal(low_pass.filter_multiplier(geometry.gen_geometry(h, w, 2), w),
                              lowpass.filter_multiplier(ogeo.gen_geometry(h, w, 2), w).reshape(h, -1))
    geom = geometry.gen_geometry(8, 16, 3)
    with pytest.raises(ValueError):
        low_pass.arakawa_1977(np.zeros((3, 8, 12)), geom)
    one = np.ones((3, 8, 1))
    assert low_pass.arakawa_1977(one, geometry.gen_geometry(8, 1, 3)) is one      # low_pass.py:58-59
