"""Altitude-dependent sensor noise (reference: sensors/models/sensor_models.py:13-22; coeff_a/b are read, never used)."""
from typing import Dict

from ...derived import _noise


class AltitudeSensorModel:
    def __init__(self, params: Dict):
        self.params = params
        self.coeff_a = params["sensor"]["model"]["coeff_a"]
        self.coeff_b = params["sensor"]["model"]["coeff_b"]

    def get_noise_variance(self, altitude) -> float:
        return _noise(altitude)
