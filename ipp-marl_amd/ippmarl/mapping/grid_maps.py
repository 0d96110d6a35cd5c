"""GridMap: cell resolution and grid dimensions derived from the config (reference: mapping/grid_maps.py:8-70)."""
from typing import Dict

from ..derived import DerivedConstants


class GridMap:
    def __init__(self, params: Dict):
        if "environment" not in params:
            raise ValueError("Cannot find environment specification in config file!")
        for key in ("x_dim", "y_dim"):
            if key not in params["environment"]:
                raise ValueError(f"Cannot find environment's {key} specification in config file!")
        self.params = params
        self._d = DerivedConstants(params)
        self.mean = None
        self.resolution_x = self.res_x
        self.resolution_y = self.res_y
        self.occupancy_matrix = None

    @property
    def x_dim(self) -> int:
        return self._d.grid_x

    @property
    def y_dim(self) -> int:
        return self._d.grid_y

    @property
    def res_x(self) -> float:
        return self._d.res_x

    @property
    def res_y(self) -> float:
        return self._d.res_y

    @property
    def num_grid_cells(self) -> int:
        return self.x_dim * self.y_dim
