"""MissionFactory (reference: missions/mission_factories.py:14-49): config -> mission object."""
from __future__ import annotations

from typing import Dict

import numpy as np

from ..utils.writers import make_writer
from .coma_mission import COMAMission
from .missions import Mission

MISSION_TYPES = ("COMA", "random", "reduced", "DeepQ")   # the types the reference maps to COMAMission (constants.py)


class MissionFactory:
    def __init__(self, params: Dict, log_dir: str = "logs", **mission_kwargs):
        self.params = params
        self.log_dir = log_dir
        self.writer = make_writer(log_dir)
        self.mission_kwargs = mission_kwargs

    @property
    def mission_type(self) -> str:
        if "missions" not in self.params["experiment"] or "type" not in self.params["experiment"]["missions"]:
            raise ValueError("Cannot find mission type specification in config file!")
        return self.params["experiment"]["missions"]["type"]

    def create_mission(self) -> Mission:
        if self.mission_type not in MISSION_TYPES:
            raise ValueError(f"'{self.mission_type}' not in list of known mission types: {MISSION_TYPES}")
        return COMAMission(self.params, self.writer, -np.inf, log_dir=self.log_dir, **self.mission_kwargs)
