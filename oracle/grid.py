"""Periodic shifts, half-point averages and one-sided gradients.

Restates coordinates.py:29-79 (2-D, axes 0=j 1=i), coordinates_3d.py:32-98
(negative axes, so one body serves [j,i] and [k,j,i]) and
coordinates_1d.py:25-53.  `unit_roll` (constants.py:85-90) is np.roll on the
magnitudes.
"""
import numpy as np


def ipj(q): return np.roll(q, -1, -1)          # coordinates.py:29-33  q[j,i+1]
def imj(q): return np.roll(q, 1, -1)           # coordinates.py:36-37  q[j,i-1]
def ijp(q): return np.roll(q, -1, -2)          # coordinates.py:40-41  q[j+1,i]
def ijm(q): return np.roll(q, 1, -2)           # coordinates.py:44-45  q[j-1,i]
def imjp(q): return imj(ijp(q))                # coordinates.py:48-49  q[j+1,i-1]
def kp(q): return np.roll(q, -1, -3)           # coordinates_3d.py:55-56
def km(q): return np.roll(q, 1, -3)            # coordinates_3d.py:59-60
def kph(q): return (q + kp(q)) / 2             # coordinates_3d.py:63-64
def kmh(q): return (q + km(q)) / 2             # coordinates_3d.py:67-68
def iph(q): return (q + ipj(q)) / 2            # coordinates.py:52-53
def imh(q): return (q + imj(q)) / 2            # coordinates.py:56-57
def jph(q): return (q + ijp(q)) / 2            # coordinates.py:60-61
def jmh(q): return (q + ijm(q)) / 2            # coordinates.py:64-65
def gradi(q, dx): return (ipj(q) - q) / dx     # coordinates.py:68-72
def gradj(q, dy): return (ijp(q) - q) / dy     # coordinates.py:75-79


# 1-D (coordinates_1d.py:25-53)
def ip(q): return np.roll(q, -1, 0)
def im(q): return np.roll(q, 1, 0)
def iph1(q): return (q + ip(q)) / 2
def imh1(q): return (q + im(q)) / 2
def div(q_h, dx): return (q_h - im(q_h)) / dx
def divu(q_h, dx): return (ip(q_h) - im(q_h)) / (2 * dx)
def gradh(q_i, dx): return (ip(q_i) - q_i) / dx


def get_total_variation(q):
    """constants.py:105-108."""
    return np.sum(np.abs(q - np.roll(q, -1, 0)))


def courant_number(p, u, dx, dt):
    """constants.py:111-112 / matsuno_c_grid.py:121-122."""
    from .constants import G
    return (np.max(u) + np.sqrt(np.mean(p) * G)) * dt / dx
