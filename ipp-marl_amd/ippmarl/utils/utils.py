"""Small shared helpers of the reference's utils/utils.py that the hot path uses."""
from collections import namedtuple

import numpy as np

# field names (and the historical tuple name) match the reference so pickled buffers stay interchangeable
TransitionCOMA = namedtuple("TransitionPPO", ("state", "observation", "action", "mask", "reward", "done", "td_target",
                                              "discounted_return"))


def compute_euclidean_distance(start: np.array, goal: np.array) -> float:
    return np.linalg.norm(np.asarray(start) - np.asarray(goal), ord=2)


def get_fixed_footprint_coordinates(footprint, footprint_clipped):
    """Offset of the clipped tile inside the unclipped-size footprint image (reference: utils/utils.py:79-98)."""
    h, w = footprint[1] - footprint[0], footprint[3] - footprint[2]
    ch, cw = footprint_clipped[1] - footprint_clipped[0], footprint_clipped[3] - footprint_clipped[2]
    yu = h - ch if footprint_clipped[0] > footprint[0] else 0
    yd = ch if footprint_clipped[1] < footprint[1] else h
    xl = w - cw if footprint_clipped[2] > footprint[2] else 0
    xr = cw if footprint_clipped[3] < footprint[3] else w
    return int(yu), int(yd), int(xl), int(xr)
