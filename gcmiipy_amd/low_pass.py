"""Drop-in for the reference's low_pass.py: the zonal Fourier damping near the poles
(`arakawa_1977`, low_pass.py:41-78), computed by the in-LDS FFT filter kernel of the 2.5-D model
(`gcm_polar_filter`), and its multiplier table."""
import numpy as np

from .dynamics import core_for
from .units import strip, attach


def filter_multiplier(geom, im):
    """S[j][n], n = 0 .. im/2 (low_pass.py:61-72): 1 for the zonal mean, and for wave number n
    1 - max(0, 1 - (1 / sin(pi n / im)) / (dy / dx_j[j])).  Host table (float64), the same values
    the library builds for its kernels."""
    dy, _ = strip(geom.dy)
    dxj, _ = strip(geom.dx_j)
    drat = (float(np.asarray(dy)) / np.asarray(dxj, dtype=np.float64)).reshape(-1, 1)
    n = np.arange(1, im // 2 + 1, dtype=np.float64)
    bysn = 1.0 / np.sin(np.pi / im * n)
    s = 1.0 - np.maximum(1.0 - bysn / drat, 0.0)
    return np.concatenate([np.ones((s.shape[0], 1)), s], axis=1)


def arakawa_1977(q, geom):
    """low_pass.py:41-78 for q of shape (H, W) or (L, H, W); W == 1 returns q (:58-59)."""
    qm, unit = strip(q)
    qm = np.asarray(qm, dtype=np.float64)
    if qm.shape[-1] == 1:
        return q
    return attach(core_for(geom).polar_filter(qm), unit)
