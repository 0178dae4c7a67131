"""Quantity handling of the drop-in functions.

The reference passes `pint.Quantity` arrays everywhere (constants.py:2-5).  The
drop-ins accept plain float64 ndarrays (SI) or anything pint-like -- an object
with `.to_base_units()` and `.m` -- convert to SI magnitudes on the way in and
re-attach the input's base units on the way out.
"""
import numpy as np


def strip(x):
    """-> (SI magnitude, unit-or-None)"""
    if hasattr(x, "to_base_units") and hasattr(x, "m"):
        b = x.to_base_units()
        return b.m, b.units
    return x, None


def scalar(x):
    m, _ = strip(x)
    return float(m)


def attach(a, unit):
    return a if unit is None else a * unit
