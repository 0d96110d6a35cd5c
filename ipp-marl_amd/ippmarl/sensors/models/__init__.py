"""Sensor noise models.  ``SensorModel`` is the interface the reference's entry scripts instantiate
(``Sensor(SensorModel(), grid_map)``); the one model that carries numbers is
:class:`ippmarl.sensors.models.sensor_models.AltitudeSensorModel` (noise level per flight altitude), whose table the HIP
kernels receive through ``DerivedConstants``."""
from abc import ABC


class SensorModel(ABC):
    """Interface only: a concrete model returns the measurement noise level at an altitude (metres)."""

    def get_noise_variance(self, altitude) -> float:  # pragma: no cover - interface
        raise NotImplementedError(f"{type(self).__name__} does not define a noise level for altitude {altitude}")
