"""Drop-in for the reference's viscosity.py (five-point Laplacian, viscosity.py:12-25) on the GPU."""
from .operators import finite_laplacian_2d, incompressible_viscosity_2d  # noqa: F401
