"""Grid geometry tables for the 2.5-D model -- host side of the drop-in.

Mirrors the reference's geometry.py call surface (`Geom`, `gen_geometry`,
`gen_square_geometry`, `manabe_sig`, `equal_sig`; geometry.py:9-182) with plain
SI float64 arrays.  The kernels read these tables, so they are evaluated with the
same expressions, in the same order, as the reference: tests compare them BIT FOR
BIT with tables captured from the reference (tests/golden/g5_geometry.npz).
"""
import math

import numpy as np

RADIUS = 6.3781e6            # constants.py:48


class Geom:
    """Attribute bag (geometry.py:9-26): sige sigt sigb dsig sig dsigv lat long dx_j dx_h dy
    area ptop heightmap, plus height/width/layers."""

    def __init__(self, height, width, layers):
        self.height, self.width, self.layers = height, width, layers
        self.ptop = 0.0
        self.dy = 0.0


def manabe_sig(s):
    return s ** 2 * (3 - 2 * s)          # geometry.py:30-31


def equal_sig(s):
    return s                             # geometry.py:34-35


def _sigma_levels(geom, layers, sig_func):
    edges = [sig_func(1 - i / (layers)) for i in range(layers + 1)]       # geometry.py:69-70
    col = lambda a: np.reshape(np.asarray(a), (len(a), 1, 1))
    geom.sige, geom.sigt, geom.sigb = col(edges), col(edges[1:]), col(edges[:-1])
    geom.dsig = geom.sigb - geom.sigt
    geom.sig = (geom.sigb + geom.sigt) / 2
    geom.dsigv = np.roll(geom.sig, -1, -3) - geom.sig                    # kp(sig) - sig


def gen_geometry(height, width, layers, sig_func=equal_sig,
                 north_edge=90, south_edge=-90, west_edge=-180, east_edge=180):
    """Lat-lon geometry, row 0 northernmost (geometry.py:38-151)."""
    geom = Geom(height, width, layers)
    _sigma_levels(geom, layers, sig_func)
    circumference = 2 * RADIUS * math.pi
    dlat = (north_edge - south_edge) / height
    dlong = (east_edge - west_edge) / width
    lat_j = np.zeros((height,))
    lat_h = np.zeros((height,))
    for i in range(height):
        lat_j[i] = north_edge - (i + 0.5) * dlat
        lat_h[i] = north_edge - (i + 1) * dlat
    long_k = np.zeros((width,))
    for i in range(width):
        long_k[i] = west_edge + (i + 0.5) * dlong
    geom.lat = lat_j.reshape((height, -1)) * (math.pi / 180)
    geom.long = long_k * (math.pi / 180)
    dx_j = np.cos(lat_j * np.pi / 180) * circumference / width
    dx_h = np.cos(lat_h * np.pi / 180) * circumference / width
    geom.dx_j = np.reshape(dx_j, (1, height, 1))
    geom.dx_h = np.reshape(dx_h, (1, height, 1))
    geom.dy = circumference / 2 / height
    geom.area = (np.roll(dx_h, 1, axis=0) + dx_h) * geom.dy * 0.5
    geom.ptop = 0 * 100.0
    geom.heightmap = np.zeros((height, width))
    return geom


def gen_square_geometry(height, width, layers, dx, dy, sig_func=equal_sig):
    """Uniform spacing (geometry.py:154-182); dx, dy in metres."""
    geom = Geom(height, width, layers)
    geom.ptop = 0 * 100.0
    _sigma_levels(geom, layers, sig_func)
    geom.lat = 0.0
    geom.long = 0.0
    geom.dx_j = np.full((1, height, 1), float(dx))
    geom.dx_h = np.full((1, height, 1), float(dx))
    geom.dy = float(dy)
    geom.heightmap = np.zeros((height, width))
    return geom


def coriolis_tables(geom):
    """cp_at_u, cp_at_v of the (disabled) Coriolis branch, dynamics.py:85-89: per latitude row,
    2 sin(lat) w and 2 sin(jph(lat)) w with w = 2 pi / day; geom.lat in radians."""
    w = 2 * math.pi / 86400.0
    lat = np.asarray(geom.lat, dtype=np.float64).reshape(-1)
    lat_h = (lat + np.roll(lat, -1)) / 2                       # jph(geom.lat), wraps like np.roll
    return 2 * np.sin(lat) * w, 2 * np.sin(lat_h) * w
