"""Mission base class (reference: missions/missions.py:5-18)."""
from typing import Dict


class Mission:
    def __init__(self, params: Dict, writer, max_mean_episode_return: float = -100):
        self.params = params
        self.writer = writer
        self.max_mean_episode_return = max_mean_episode_return

    def execute(self):
        raise NotImplementedError("Planning mission does not implement 'execute' function!")
