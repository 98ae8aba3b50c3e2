"""ev-charge.py's charging-curve model (reference ev-charge.py:17-18), importable."""
import numpy as np


def charge(t, P0=0, P_max=189.0, t_max=3.5, a=6.9077):
    """Charge status (kWh) after t hours: P_max (1 - exp(-a t / t_max)) + P0."""
    return P_max * (1 - np.exp(-a * np.asarray(t, float) / t_max)) + P0
