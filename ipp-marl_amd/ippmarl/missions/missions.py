"""What every mission shares: the config, a scalar sink and the best running mean return seen so far.  A concrete mission
(``coma_mission.COMAMission``) supplies ``execute`` (the reference keeps the same three attributes on its base class,
missions/missions.py:5-18)."""
from __future__ import annotations

from typing import Dict


class Mission:
    def __init__(self, params: Dict, writer=None, max_mean_episode_return: float = -100.0):
        self.params, self.writer = params, writer
        self.max_mean_episode_return = float(max_mean_episode_return)

    def execute(self):
        """Runs the mission and returns its figure of merit."""
        raise NotImplementedError(f"{type(self).__name__} has no execute()")
