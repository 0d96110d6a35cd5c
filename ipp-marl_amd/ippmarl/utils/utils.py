"""Small shared helpers of the reference's utils/utils.py that the hot path uses."""
from collections import namedtuple

import numpy as np

# field names (and the historical tuple name) match the reference so pickled buffers stay interchangeable
TransitionCOMA = namedtuple("TransitionPPO", ("state", "observation", "action", "mask", "reward", "done", "td_target",
                                              "discounted_return"))


def compute_euclidean_distance(start: np.array, goal: np.array) -> float:
    return np.linalg.norm(np.asarray(start) - np.asarray(goal), ord=2)


def get_wrmse(map_state, map_simulation, params=None):
    """F1 score of the target class for the map thresholded at p > 0.5 against the ground truth (reference: utils/utils.py:43-76;
    despite its name the function returns ``f1_score(truth, rounded, average=None)[1]``).  The counts come from
    ``ippm_f1_counts`` on the device; p - 0.5 is handed over as the score, so the threshold is exactly the reference's."""
    import torch

    from .. import _ffi
    from .._engine import engine_for_grid, scratch_engine
    state = np.asarray(map_state)
    eng = scratch_engine(params) if params is not None else (engine_for_grid(state.shape) or scratch_engine(_default_params()))
    env = eng.env
    d = env.d
    if state.shape != (d.grid_x, d.grid_y):
        raise _ffi.IppmError(f"get_wrmse: map of shape {state.shape}, the configuration's grid is {(d.grid_x, d.grid_y)}")
    score = torch.from_numpy(np.ascontiguousarray(state.astype(np.float64) - 0.5, dtype=np.float32)).to(env.device)
    # (float32(p - 0.5) keeps the sign of p - 0.5 for every float32 / float64 p)
    truth = torch.from_numpy(d.pack_truth(np.asarray(map_simulation) > 0.5)).to(env.device).view(1, -1)
    counts = torch.zeros(1, 3, dtype=torch.int64, device=env.device)
    env.ctx.call("ippm_f1_counts", _ffi.ptr(score), _ffi.ptr(truth), 1, 0.0, _ffi.ptr(counts), 1, env.stream)
    tp, fp, fn = (int(v) for v in counts[0].cpu())
    return 2 * tp / (2 * tp + fp + fn) if (2 * tp + fp + fn) > 0 else 0.0


_DEFAULT_PARAMS = None


def _default_params():
    global _DEFAULT_PARAMS
    if _DEFAULT_PARAMS is None:
        from ..params import load_params
        _DEFAULT_PARAMS = load_params()
    return _DEFAULT_PARAMS


def get_fixed_footprint_coordinates(footprint, footprint_clipped):
    """Offset of the clipped tile inside the unclipped-size footprint image (reference: utils/utils.py:79-98)."""
    h, w = footprint[1] - footprint[0], footprint[3] - footprint[2]
    ch, cw = footprint_clipped[1] - footprint_clipped[0], footprint_clipped[3] - footprint_clipped[2]
    yu = h - ch if footprint_clipped[0] > footprint[0] else 0
    yd = ch if footprint_clipped[1] < footprint[1] else h
    xl = w - cw if footprint_clipped[2] > footprint[2] else 0
    xr = cw if footprint_clipped[3] < footprint[3] else w
    return int(yu), int(yd), int(xl), int(xr)
