"""Camera footprint projection through the K2 kernel (reference: sensors/cameras.py:31-79)."""
from typing import Dict, Tuple

import numpy as np
import torch

from .. import _ffi
from .._engine import scratch_engine
from ..mapping.grid_maps import GridMap
from . import Sensor


class Camera(Sensor):
    def __init__(self, params: Dict, sensor_model, grid_map: GridMap):
        super().__init__(sensor_model, grid_map)
        self.params = params
        self.grid_map = GridMap(params)

    @property
    def angle_x(self) -> float:
        return self.params["sensor"]["field_of_view"]["angle_x"]

    @property
    def angle_y(self) -> float:
        return self.params["sensor"]["field_of_view"]["angle_y"]

    def field_of_view_range(self, height: float) -> Tuple[float, float]:
        return (2 * height * np.tan(0.5 * np.radians(self.angle_x)), 2 * height * np.tan(0.5 * np.radians(self.angle_y)))

    def project_field_of_view(self, position: np.array, res_x=None, res_y=None):
        """-> ([yu,yd,xl,xr] unclipped, clipped).  The resolutions are derived from the config (the arguments exist for
        signature compatibility)."""
        env = scratch_engine(self.params).env
        n = env.d.n_agents
        pos = torch.zeros(1, n, 3, dtype=torch.int32)
        pos[0, :, 2] = env.d.min_altitude
        pos[0, 0] = torch.as_tensor(np.asarray(position, dtype=np.int32))
        pos = pos.to(env.device)
        rect, full = env.footprints(pos)
        return [int(v) for v in full[0, 0].cpu()], [int(v) for v in rect[0, 0].cpu()]
