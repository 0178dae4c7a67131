"""Headless twin of the reference's 1-D harness `just_units.run_1d_with_ft` (just_units.py:298-340):
`state = ft(**state)` for `steps` steps under the total-variation watch.  The Matplotlib window is
left out; the watch itself -- including the early `return False` the 1-D harness still has, unlike
two_d.run_2d_with_ft (two_d.py:338) -- is the reference's.  The reductions run on the GPU
(constants.get_total_variation -> gcm_array_stats); a NaN anywhere in the watched field makes the
total variation NaN, which is the reference's `np.isnan(...).any()` exit."""
import math

from .constants import get_total_variation
from .units import strip


def run_1d_with_ft(initial_conditions, ft, steps=400, display_key="q", variation_key="q", history=None):
    """just_units.py:298-340 -> True, or False as soon as the total variation of `variation_key` has
    grown by more than 1000 over its initial value or the field holds a NaN (:327-332).  Pass a list
    as `history` to receive the variation series; the last state is `run_1d_with_ft.last_state`."""
    current = initial_conditions
    initial_variation = get_total_variation(strip(current[variation_key])[0])
    if history is not None:
        history.append(initial_variation)
    for _ in range(steps):
        current = ft(**current)
        run_1d_with_ft.last_state = current
        v = get_total_variation(strip(current[variation_key])[0])
        if history is not None:
            history.append(v)
        if initial_variation + 1000 < v or math.isnan(v):
            return False
    run_1d_with_ft.last_state = current
    return True
