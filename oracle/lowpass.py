"""Zonal Fourier damping near the poles (reference low_pass.py:41-78)."""
import numpy as np


def filter_multiplier(geom, im):
    """low_pass.py:61-72: S[.,j,n], S[j,0] = 1,
    S[j,n>=1] = 1 - max(0, 1 - (1/sin(pi n/W)) / (dy/dx_j[j]))."""
    drat = geom.dy / geom.dx_j
    nmax = im / 2
    bysn = 1 / np.sin(np.pi / im * np.arange(1, nmax + 1))
    sm = 1 - bysn / drat
    smmz = 1 - np.maximum(sm, np.zeros_like(sm))
    return np.insert(smmz, 0, 1, -1)


def arakawa_1977(q, geom):
    """low_pass.py:41-78.  Even W only; W == 1 is the identity (:58-59)."""
    im = q.shape[-1]
    if im == 1:
        return q
    smmz = filter_multiplier(geom, im)
    f_q = np.fft.rfft(q)
    f_q_f = f_q * smmz
    return np.fft.irfft(f_q_f)
