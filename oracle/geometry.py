"""Grid geometry tables (reference geometry.py:9-182).  All SI magnitudes.

These tables are the "geometry" items that must be BIT-EXACT between the
reference and the build; gcmiipy_amd.geometry evaluates the same expressions
and tests compare the two bit for bit.
"""
import math

import numpy as np

from .constants import radius
from .grid import kp


class Geom:
    """geometry.py:9-26 attribute bag."""

    def __init__(self, height, width, layers):
        self.height = height
        self.width = width
        self.layers = layers


def manabe_sig(s):
    """geometry.py:30-31."""
    return s ** 2 * (3 - 2 * s)


def equal_sig(s):
    """geometry.py:34-35."""
    return s


def _sigma(geom, layers, sig_func):
    mysig = []
    for i in range(layers + 1):
        mysig.append(sig_func(1 - i / (layers)))

    def rs(arr):
        return np.reshape(arr, (arr.shape[0], 1, 1))

    geom.sige = rs(np.asarray(mysig))
    geom.sigt = rs(np.asarray(mysig[1:]))
    geom.sigb = rs(np.asarray(mysig[:-1]))
    geom.dsig = geom.sigb - geom.sigt
    geom.sig = (geom.sigb + geom.sigt) / 2
    geom.dsigv = kp(geom.sig) - geom.sig


def gen_geometry(height, width, layers, sig_func=equal_sig,
                 north_edge=90, south_edge=-90, west_edge=-180, east_edge=180):
    """geometry.py:38-151 (the seven print()s are not reproduced)."""
    geom = Geom(height, width, layers)
    _sigma(geom, layers, sig_func)                                  # :67-85

    circumference = 2 * radius * math.pi                            # :88
    lat_j = np.zeros((height,))
    lat_h = np.zeros((height,))
    dlat = (north_edge - south_edge) / height
    dlong = (east_edge - west_edge) / width
    for i in range(height):
        lat_j[i] = north_edge - (i + 0.5) * dlat                    # :99
        lat_h[i] = north_edge - (i + 1) * dlat                      # :100
    long_k = np.zeros((width,))
    for i in range(width):
        long_k[i] = west_edge + (i + 0.5) * dlong

    geom.lat = lat_j.reshape((height, -1)) * (math.pi / 180)        # :107 (radians)
    geom.long = long_k * (math.pi / 180)

    cos_j = np.cos(lat_j * np.pi / 180)                             # :110
    cos_h = np.cos(lat_h * np.pi / 180)                             # :112
    dx_j = cos_j * circumference / width                            # :114
    dx_h = cos_h * circumference / width                            # :115

    geom.dx_j = np.reshape(dx_j, (1, height, 1))                    # :136
    geom.dx_h = np.reshape(dx_h, (1, height, 1))                    # :137
    geom.dy = circumference / 2 / height                            # :138
    geom.area = (np.roll(dx_h, 1, axis=0) + dx_h) * geom.dy * 0.5   # :141-142
    geom.ptop = 0 * 100.0                                           # :147 (0 hPa)
    geom.heightmap = np.zeros((height, width))                      # :149
    return geom


def gen_square_geometry(height, width, layers, dx, dy, sig_func=equal_sig):
    """geometry.py:154-182."""
    geom = Geom(height, width, layers)
    geom.ptop = 0 * 100.0
    _sigma(geom, layers, sig_func)
    geom.lat = 0.0
    geom.long = 0.0
    geom.dx_j = np.full((1, height, 1), dx) * 1.0
    geom.dx_h = np.full((1, height, 1), dx) * 1.0
    geom.dy = dy
    geom.heightmap = np.zeros((height, width))
    return geom
