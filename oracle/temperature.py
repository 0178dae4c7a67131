"""theta <-> T (reference temperature.py:7-24)."""
from .constants import P0, kappa, Rd


def to_true_temp(t, p):
    """temperature.py:7-12."""
    if hasattr(t, "shape"):
        assert t.shape == p.shape
    return t / ((P0 / p) ** kappa)


def to_potential_temp(tt, p):
    """temperature.py:15-19."""
    if hasattr(tt, "shape"):
        assert tt.shape == p.shape
    return tt * ((P0 / p) ** kappa)


def to_density(tt, p):
    """temperature.py:22-24."""
    return p / (Rd * tt)
