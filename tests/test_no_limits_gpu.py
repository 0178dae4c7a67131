"""GPU parity of the 1-D model drop-in gcmiipy_amd.no_limits (BASELINE configs[0]; reference
no_limits.py:115-152) vs the golden vectors G9 (the reference itself run on its own initial state)
and vs the oracle on seeded states, ragged lengths included."""
import numpy as np
import pytest

from conftest import golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_matsuno_1d_vs_golden():
    from gcmiipy_amd import no_limits
    d = golden("g9_oned")
    st = tuple(d[k + "0"] for k in "putq")
    s2 = st
    for _ in range(2):                                             # the host-visible call, step by step
        s2 = no_limits.matsuno_timestep(*s2, 900.0, 70000.0)
    for k, x in zip("putq", s2):
        assert rel_err(x, d[k + "2_900"]) < TOL, k
    got = no_limits.run(*st, float(d["dt"]), float(d["dx"]), 10)   # resident on the device
    for k, x in zip("putq", got):
        assert rel_err(x, d[k + "10"]) < TOL, k


@pytest.mark.parametrize("n", [1, 2, 3, 64, 257, 100000])
def test_half_and_full_step_vs_oracle(n):
    from gcmiipy_amd import no_limits
    from oracle import oned
    rng = np.random.default_rng(n)
    p = 90000.0 + 5000.0 * rng.random(n)
    u = 20.0 * rng.standard_normal(n)
    t = 300.0 + 10.0 * rng.random(n)
    q = rng.random(n)
    sp, su, st, sq = p + rng.random(n), u + rng.random(n), t + rng.random(n), q + 0.01 * rng.random(n)
    dt, dx = 60.0, 50000.0
    for got, want, k in zip(no_limits.half_timestep(p, u, t, q, sp, su, st, sq, dt, dx),
                            oned.half_timestep(p, u, t, q, sp, su, st, sq, dt, dx), "putq"):
        assert rel_err(got, want) < TOL, k
    a = (p, u, t, q)
    for _ in range(3):
        a = oned.matsuno_timestep(*a, dt, dx)
    for got, want, k in zip(no_limits.run(p, u, t, q, dt, dx, 3), a, "putq"):
        assert rel_err(got, want) < TOL, k
    for got, want in zip(no_limits.run(p, u, t, q, dt, dx, 0), (p, u, t, q)):
        assert np.array_equal(got, want)


def test_units_and_errors():
    from gcmiipy_amd import no_limits

    class FakeQ:                                   # pint-like: .to_base_units(), .m, .units
        def __init__(self, m, factor):
            self._m, self._f = m, factor
        def to_base_units(self):
            return FakeQ(self._m * self._f, 1.0)
        m = property(lambda s: s._m)
        units = property(lambda s: 1.0)

    d = golden("g9_oned")
    p, u, t, q = (d[k + "0"] for k in "putq")
    want = no_limits.matsuno_timestep(p, u, t, q, 900.0, 70000.0)
    got = no_limits.matsuno_timestep(FakeQ(p / 100.0, 100.0), u, t, q, FakeQ(15.0, 60.0), FakeQ(70.0, 1000.0))
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        no_limits.matsuno_timestep(p, u[:-1], t, q, 900.0, 70000.0)
    with pytest.raises(ValueError):
        no_limits.matsuno_timestep(p.reshape(2, -1), u.reshape(2, -1), t.reshape(2, -1), q.reshape(2, -1), 900.0, 70000.0)
    with pytest.raises(ValueError):
        no_limits.matsuno_timestep(p, u, t, q, 900.0, 0.0)
