class SensorModel:
    def get_noise_variance(self, altitude) -> float:
        raise NotImplementedError("Sensor has no noise variance function implemented")
