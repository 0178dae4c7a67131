"""Physical constants, SI magnitudes (reference constants.py:10-78)."""
R = 8.3145                      # constants.py:10
Md = 28.97 * 1e-3               # constants.py:13 (g/mol -> kg/mol)
Rd = 287.0                      # constants.py:16
Cp = 1004.0                     # constants.py:22
kappa = Rd / Cp                 # constants.py:28
P0 = 100000.0                   # constants.py:31
standard_pressure = 101325.0    # constants.py:37
standard_temperature = 273.16   # constants.py:38
G = 9.8                         # constants.py:45
radius = 6.3781e6               # constants.py:48
mu_air = 18.5 * 1e-6            # constants.py:51 (uPa s -> Pa s)
Rv = 461.0                      # constants.py:78
