"""Weighted-entropy maps on the GPU (reference: utils/state.py:14-121): the functions the reference's baselines and
deployment scripts import by name (IG_baseline.py:28, coma_test.py:25).

Same signatures, return tuples and mutation conventions: ``get_shannon_entropy`` clips its argument IN PLACE
(utils/state.py:118-121), so the ``grid_map`` handed back by ``get_w_entropy_map`` is the clipped copy.  The arithmetic runs
in two kernels of libippmarl.so: ``ippm_area_resize`` (cv2.resize(..., INTER_AREA) to the 11 x 11 lattice, exact area
average) and ``ippm_entropy_maps`` (clip, Shannon entropy, class weights, product, per element).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _ffi
from .._engine import scratch_engine


def _ctx_of(agent_state_space):
    params = getattr(agent_state_space, "params", None)
    if params is None:
        from ..params import load_params
        params = load_params()
    env = scratch_engine(params).env
    return env


def _resize(env, src: np.ndarray, space_dim) -> np.ndarray:
    """cv2.resize(src, (space_dim[1], space_dim[0]), interpolation=cv2.INTER_AREA) on the device."""
    if tuple(int(v) for v in space_dim[:2]) != (_ffi.FEAT, _ffi.FEAT):
        raise _ffi.IppmError("the device resize targets the 11 x 11 lattice the networks are wired for (actor/network.py:19-21)")
    a = np.ascontiguousarray(src, dtype=np.float32)
    t = torch.from_numpy(a).to(env.device)
    dst = torch.empty(_ffi.FEAT, _ffi.FEAT, dtype=torch.float32, device=env.device)
    scratch = torch.empty(_ffi.FEAT * _ffi.FEAT, dtype=torch.float64, device=env.device)
    env.ctx.call("ippm_area_resize", _ffi.ptr(t), a.shape[0], a.shape[1], _ffi.ptr(dst), _ffi.ptr(scratch), 1, env.stream)
    return dst.cpu().numpy().astype(src.dtype if np.issubdtype(np.asarray(src).dtype, np.floating) else np.float32)


def _entropy_maps(env, prob: np.ndarray, target: np.ndarray = None):
    """-> (weightings f32, se f32, clipped f32) of a probability array; weights from ``target`` (default: the array)."""
    p = torch.from_numpy(np.ascontiguousarray(prob, dtype=np.float32)).to(env.device)
    tg = None if target is None else torch.from_numpy(np.ascontiguousarray(target, dtype=np.float32)).to(env.device)
    w, se, grid = torch.empty_like(p), torch.empty_like(p), torch.empty_like(p)
    env.ctx.call("ippm_entropy_maps", _ffi.ptr(p), _ffi.ptr(tg), None, _ffi.ptr(w), _ffi.ptr(se), _ffi.ptr(grid), p.numel(),
                 env.stream)
    return w.cpu().numpy(), se.cpu().numpy(), grid.cpu().numpy()


def get_shannon_entropy(p: np.ndarray, agent_state_space=None) -> np.ndarray:
    """-p log2 p - (1-p) log2 (1-p) after clipping ``p`` to [1e-4, 0.9999] in place (utils/state.py:118-121)."""
    env = _ctx_of(agent_state_space)
    _, se, grid = _entropy_maps(env, p)
    p[...] = grid.astype(p.dtype)
    return se.astype(p.dtype)


def calculate_w_entropy(grid_map: np.ndarray, map_footprint, simulated_map, observability: str, agent_state_space):
    """-> (w_entropy_map, weightings, se, w_entropy_map_footprint, grid_map) (utils/state.py:53-115).  Weights: 1 where the
    target exceeds 0.501, 0 below 0.499, 0.5 between (class_weighting [0, 1]); target = the ground truth for "eval", the map
    itself otherwise."""
    env = _ctx_of(agent_state_space)
    target = simulated_map if observability == "eval" else None
    w, se, clipped = _entropy_maps(env, grid_map, target)
    tdtype = np.asarray(simulated_map).dtype if observability == "eval" else grid_map.dtype
    weightings = w.astype(tdtype)
    grid_map[...] = clipped.astype(grid_map.dtype)            # the reference's in-place clip
    se = se.astype(grid_map.dtype)
    w_entropy_map = weightings * se
    w_entropy_map_footprint = None
    if observability == "actor":
        wf, sef, clipf = _entropy_maps(env, map_footprint)
        map_footprint[...] = clipf.astype(map_footprint.dtype)
        w_entropy_map_footprint = wf.astype(map_footprint.dtype) * sef.astype(map_footprint.dtype)
    return w_entropy_map, weightings, se, w_entropy_map_footprint, grid_map


def get_w_entropy_map(map_footprint, local_map, simulated_map, observability: str, agent_state_space):
    """utils/state.py:14-50: "reward" / "eval" work at full resolution, every other mode ("actor", "global", ...) first
    resizes the map (and the ground truth; for "actor" also the footprint image) to the lattice."""
    env = _ctx_of(agent_state_space)
    if observability not in ("reward", "eval"):
        dim = agent_state_space.space_dim
        grid_map = _resize(env, np.asarray(local_map), dim)
        if observability == "actor":
            map_footprint = _resize(env, np.asarray(map_footprint), dim)
        simulated_map = _resize(env, np.asarray(simulated_map), dim)
    else:
        grid_map = np.array(local_map, copy=True)
    return calculate_w_entropy(grid_map, map_footprint, simulated_map, observability, agent_state_space)
