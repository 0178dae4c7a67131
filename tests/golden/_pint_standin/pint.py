"""Unit-bookkeeping stand-in for the `pint` package (absent from this image).

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py, inside the
build container, so that the *unmodified* reference modules under
/root/reference can be imported and executed to produce golden vectors.  It
never travels into the product path and nothing under gcmiipy_amd/ imports it.

Design: a Quantity is a plain wrapper (NOT an ndarray subclass: NumPy's
temporary elision would silently drop a subclass on arrays >= 256 KiB) around
float64 magnitudes that are converted to SI base units when the quantity is
constructed (``x * units.km`` multiplies by 1000.0).  After construction every
operator and NumPy function is forwarded to NumPy on the raw magnitudes in the
order the calling code wrote it, so the arithmetic that is executed is the
reference's own NumPy arithmetic.  Dimension tracking is not reproduced:
``.u`` / ``.units`` are the scalar 1.0.
"""
import math
import operator

import numpy as np

_SI = dict(
    m=1.0, meter=1.0, km=1000.0, s=1.0, second=1.0, minute=60.0, hour=3600.0,
    hours=3600.0, day=86400.0, days=86400.0, K=1.0, kelvin=1.0, Kelvin=1.0,
    Pa=1.0, hPa=100.0, kPa=1000.0, uPa=1e-6, J=1.0, W=1.0, watt=1.0,
    kW=1000.0, mol=1.0, g=1e-3, gram=1e-3, grams=1e-3, kg=1.0,
    degrees=math.pi / 180, degree=math.pi / 180, radian=1.0,
    dimensionless=1.0,
)


_CELSIUS = None  # sentinel unit, bound to a Quantity once the class exists


def _raw(x):
    return x._v if isinstance(x, Quantity) else x


def _raw_tree(x):
    if isinstance(x, Quantity):
        return x._v
    if isinstance(x, (list, tuple)):
        return type(x)(_raw_tree(i) for i in x)
    if isinstance(x, dict):
        return {k: _raw_tree(v) for k, v in x.items()}
    return x


def _wrap(x):
    if isinstance(x, tuple):
        return tuple(_wrap(i) for i in x)
    if isinstance(x, np.ndarray) and x.dtype.kind == "f":
        return Quantity(x)
    if isinstance(x, (float, np.floating)):
        return Quantity(x)
    return x


class Quantity:
    __array_priority__ = 1000
    __hash__ = None

    def __init__(self, value, unit=None):
        value = _raw(value)
        if unit is not None and unit is _CELSIUS:
            value = np.asarray(value, dtype=np.float64) + 273.15
        elif unit is not None:
            value = value * _raw(unit)
        self._v = value

    # --- pint attribute surface -----------------------------------------
    @property
    def m(self):
        return self._v

    magnitude = m

    @property
    def u(self):
        return Quantity(1.0)

    units = u

    # like pint, these forward to the magnitude: a Python-float quantity has
    # no .shape (temperature.py:8 relies on that via hasattr)
    @property
    def shape(self):
        return self._v.shape

    @property
    def ndim(self):
        return self._v.ndim

    @property
    def dtype(self):
        return self._v.dtype

    def to_base_units(self):
        return self

    def to(self, other):
        if other is not None and other is _CELSIUS:
            return Quantity(self._v - 273.15)
        return Quantity(self._v / _raw(other))

    def flatten(self):
        return Quantity(np.asarray(self._v).flatten())

    def any(self, *a, **k):
        return np.asarray(self._v).any(*a, **k)

    def all(self, *a, **k):
        return np.asarray(self._v).all(*a, **k)

    def copy(self):
        return Quantity(np.copy(self._v))

    # --- numpy protocol ----------------------------------------------------
    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._v, dtype=dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if "out" in kwargs:
            kwargs["out"] = tuple(_raw(o) for o in kwargs["out"])
        if "where" in kwargs:
            kwargs["where"] = _raw(kwargs["where"])
        res = getattr(ufunc, method)(*[_raw(i) for i in inputs], **kwargs)
        return _wrap(res)

    def __array_function__(self, func, types, args, kwargs):
        res = func(*_raw_tree(args), **_raw_tree(kwargs))
        return _wrap(res)

    # --- container protocol -------------------------------------------------
    def __len__(self):
        return len(self._v)

    def __iter__(self):
        for item in self._v:
            yield _wrap(item) if isinstance(item, (np.ndarray, float, np.floating)) else item

    def __getitem__(self, idx):
        return _wrap(self._v[_raw_tree(idx)])

    def __setitem__(self, idx, value):
        self._v[_raw_tree(idx)] = _raw(value)

    # --- scalar conversions / formatting -----------------------------------
    def __float__(self):
        return float(self._v)

    def __int__(self):
        return int(self._v)

    def __index__(self):
        return operator.index(self._v)

    def __bool__(self):
        return bool(self._v)

    def __repr__(self):
        return "Q(%r)" % (self._v,)

    def __str__(self):
        return str(self._v)

    def __format__(self, spec):
        return format(self._v, spec)

    def __neg__(self):
        return Quantity(-self._v)

    def __pos__(self):
        return self

    def __abs__(self):
        return Quantity(abs(self._v))

    def __round__(self, n=None):
        return Quantity(round(float(self._v), n))


def _binary(name, op, wrap=True):
    def fwd(self, other):
        res = op(self._v, _raw(other))
        return _wrap(res) if wrap else res

    def rev(self, other):
        res = op(_raw(other), self._v)
        return _wrap(res) if wrap else res

    def inplace(self, other):
        if isinstance(self._v, np.ndarray):
            # same in-place ufunc NumPy would run for `a op= b`
            self._v = getattr(operator, "i" + name)(self._v, _raw(other))
        else:
            self._v = op(self._v, _raw(other))
        return self

    setattr(Quantity, "__%s__" % name, fwd)
    setattr(Quantity, "__r%s__" % name, rev)
    setattr(Quantity, "__i%s__" % name, inplace)


for _n, _o in (("add", operator.add), ("sub", operator.sub), ("mul", operator.mul),
               ("truediv", operator.truediv), ("floordiv", operator.floordiv),
               ("mod", operator.mod), ("pow", operator.pow)):
    _binary(_n, _o)

for _n, _o in (("lt", operator.lt), ("le", operator.le), ("gt", operator.gt),
               ("ge", operator.ge), ("eq", operator.eq), ("ne", operator.ne)):
    setattr(Quantity, "__%s__" % _n,
            (lambda o: lambda self, other: o(self._v, _raw(other)))(_o))

_CELSIUS = Quantity(1.0)


class UnitRegistry:
    Quantity = Quantity

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name == "celsius":
            return _CELSIUS
        try:
            return Quantity(_SI[name])
        except KeyError:
            raise AttributeError("pint stand-in: unknown unit %r" % name)
