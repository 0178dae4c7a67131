"""Drop-in for the reference's temperature.py (temperature.py:7-24) on the GPU: the Exner-function
conversions between potential and true temperature, and the ideal-gas density."""
from .operators import to_true_temp, to_potential_temp, to_density  # noqa: F401
