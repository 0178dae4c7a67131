"""GPU parity of GCM_PE2D (no_limits_2d.py) vs the golden vectors G10 and the oracle."""
import numpy as np
import pytest

from conftest import golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_pe2d_golden_steps_and_dropin():
    import gcmiipy_amd as g
    from gcmiipy_amd import no_limits_2d
    d = golden("g10_pe2d")
    st = [d[k + "0"] for k in "puvtq"]
    dx, dt = float(d["dx"]), float(d["dt"])
    c = g.Core(g._lib.PE2D, 32, 16, dx=dx)
    c.set_state(*st)
    done = 0
    for n in (1, 5):
        c.step(n - done, dt)
        done = n
        for k, x in zip("puvtq", c.get_state()):
            assert rel_err(x, d["%s%d" % (k, n)]) < TOL, (k, n)
    c.close()
    one = no_limits_2d.matsuno_timestep(*st, dt, dx)
    for k, x in zip("puvtq", one):
        assert rel_err(x, d[k + "1"]) < TOL, k
    assert np.array_equal(one[4], st[4])                       # q passes through, :126
    # half_timestep(base; base) then (base; star) == matsuno_timestep
    star = no_limits_2d.half_timestep(*st, *st, dt, dx)
    two = no_limits_2d.half_timestep(*st, *star, dt, dx)
    for k, x, y in zip("puvtq", two, one):
        assert rel_err(x, y) < 1e-14, k


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (1, 9), (7, 1), (33, 130)])
def test_pe2d_shapes_vs_oracle(shape):
    import gcmiipy_amd as g
    from oracle import pe2d
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    p = 101325 + 10 * rng.standard_normal(shape)
    u, v = rng.standard_normal(shape), rng.standard_normal(shape)
    t = 300 + rng.standard_normal(shape)
    q = rng.random(shape)
    want = (p, u, v, t, q)
    for _ in range(3):
        want = pe2d.matsuno_timestep(*want, 100.0, 100e3)
    c = g.Core(g._lib.PE2D, shape[1], shape[0], dx=100e3)
    c.set_state(p, u, v, t, q)
    c.step(3, 100.0)
    for k, x, y in zip("puvtq", c.get_state(), want):
        assert rel_err(x, y) < TOL, (k, rel_err(x, y))
    c.close()
