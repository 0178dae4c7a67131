"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/gcmcore.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gcmcore.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gcm_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import gcmiipy_amd
    from gcmiipy_amd import _lib
    names = _declared()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libgcmcore.so lacks %s" % n
        assert n in _lib.SYMBOLS, "ctypes binding lacks %s" % n
    assert sorted(_lib.SYMBOLS) == names
    assert _lib.lib.gcm_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in _lib.lib.gcm_build_info()


def test_config_struct_layout_matches_header():
    """field order/types of the ctypes mirror follow the header's gcm_config."""
    from gcmiipy_amd import _lib
    src = open(os.path.join(ROOT, "include", "gcmcore.h")).read()
    body = src[src.index("typedef struct {"):src.index("} gcm_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:int32_t|double|const double \*|void \*)\s*\*?(\w+);", body)
    assert fields == [f[0] for f in _lib.Config._fields_]
    assert ctypes.sizeof(_lib.Config) == 16 * 4 + 3 * 8 + 10 * 8      # 15 int32 + 4 bytes padding


def test_no_cpu_fallback_without_device():
    import gcmiipy_amd as g
    if g.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(g.GcmError, match="no CPU fallback"):
        g.Core(g._lib.SW2D, 32, 16, dx=1.0)
    from gcmiipy_amd.matsuno_c_grid import matsumo_scheme
    z = np.zeros((4, 4))
    with pytest.raises(g.GcmError):
        matsumo_scheme(z, z, z + 1, 1.0, 1.0)


def test_operator_drop_ins_fail_loudly_without_device():
    """every per-operator entry point (no handle: host arrays in, host arrays out) refuses to run
    without a gfx950 device -- no NumPy fallback hides behind the reference's function names"""
    from gcmiipy_amd import _lib
    if _lib.lib.gcm_device_count() != 0:
        pytest.skip("a HIP device is present")
    from gcmiipy_amd import (two_d, flux_limiter, no_limits, matsuno_c_grid, matsumo_temp, viscosity, temperature,
                             dynamics, no_limits_2d, geometry, low_pass)
    from gcmiipy_amd.core import GcmError
    a = np.ones((4, 6))
    a3 = np.ones((2, 4, 6))
    geom = geometry.gen_geometry(4, 6, 2)
    calls = [lambda: two_d.gradient(a, (1.0, 1.0), 0), lambda: two_d.pgf_one_d(1.0, 1.0, a),
             lambda: flux_limiter.calc_r(np.ones(8)), lambda: no_limits.matsuno_timestep(*[np.ones(8)] * 4, 1.0, 1.0),
             lambda: matsuno_c_grid.advection_of_velocity_u(a, a, 1.0), lambda: matsumo_temp.density_from(a, a),
             lambda: viscosity.finite_laplacian_2d(a, 1.0), lambda: temperature.to_true_temp(a, a),
             lambda: dynamics.aflux(a3, a3, geom), lambda: dynamics.calc_pu(a, a3),
             lambda: no_limits_2d.advec_m(a, a, a, 1.0), lambda: low_pass.arakawa_1977(a3, geom)]
    for f in calls:
        with pytest.raises(GcmError, match="no HIP device"):
            f()


def test_argument_validation_mirrors_reference_asserts():
    import gcmiipy_amd as g
    from gcmiipy_amd.matsuno_c_grid import matsumo_scheme
    from gcmiipy_amd.matsumo_temp import matsumo_scheme as ms_t
    z = np.zeros((4, 4))
    with pytest.raises(ValueError):
        matsumo_scheme(z, np.zeros((4, 5)), z, 1.0, 1.0)      # shape mismatch
    with pytest.raises(ValueError):
        matsumo_scheme(np.zeros(4), np.zeros(4), np.zeros(4), 1.0, 1.0)
    with pytest.raises(ValueError):
        ms_t(z, z, z, np.zeros((3, 4)), 1.0, 1.0)
    with pytest.raises(ValueError):
        g.Core(g._lib.SW2D, 0, 16, dx=1.0)                    # rejected before any device use
    with pytest.raises(ValueError):
        g.Core(g._lib.SW2D, 8, 8, dx=1.0, nranks=2, rank=2)


def test_product_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under gcmiipy_amd/ may import it."""
    pkg = os.path.join(ROOT, "gcmiipy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "/root/reference" not in txt, f


def test_unit_stripping_roundtrip():
    from gcmiipy_amd.units import strip, attach, scalar

    class FakeQ:                                   # pint-like: .to_base_units(), .m, .units
        def __init__(self, m, factor):
            self._m, self._f = m, factor
        def to_base_units(self):
            return FakeQ(self._m * self._f, 1.0)
        m = property(lambda s: s._m)
        units = property(lambda s: 1.0)

    m, u = strip(FakeQ(np.ones(3) * 300.0, 1000.0))
    assert np.array_equal(m, np.full(3, 300e3)) and u == 1.0
    assert scalar(FakeQ(2.0, 60.0)) == 120.0
    assert strip(np.ones(2))[1] is None
    assert attach(np.ones(2), None).shape == (2,)


def test_physics_and_exchange_struct_layouts_match_header():
    """the ctypes mirrors of gcm_physics and gcm_exchange follow the header field for field"""
    from gcmiipy_amd import _lib
    src = open(os.path.join(ROOT, "include", "gcmcore.h")).read()

    def fields_of(name):
        end = src.index("} %s;" % name)
        body = src[src.rindex("typedef struct {", 0, end):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.replace("typedef struct {", "").strip()
            if not decl:
                continue
            names = decl.split(",")
            first = names[0].split()[-1].lstrip("*")
            out.append(first)
            out += [n.strip().lstrip("*") for n in names[1:]]
        return out

    assert fields_of("gcm_physics") == [f[0] for f in _lib.Physics._fields_]
    assert ctypes.sizeof(_lib.Physics) == 4 * 8 + 2 * 8
    assert fields_of("gcm_exchange") == [f[0] for f in _lib.Exchange._fields_]
    assert ctypes.sizeof(_lib.Exchange) == 8 + 2 * 4 + 4 * 8 + 4 * 8
