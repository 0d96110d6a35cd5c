"""Per-episode device engine behind the drop-in classes (Mapping / Agent / COMAWrapper).

One ``EpisodeEngine`` = a VecEnv with a single environment whose device tensors are the source of truth; the
reference-shaped objects are views that upload/download NumPy arrays at the API boundary (probabilities outside,
log-odds inside).  Throughput work uses VecEnv / COMATrainer directly; this layer exists so that code written against
the reference's object surface runs unchanged on the HIP kernels.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import _ffi
from .vec_env import VecEnv

_ENGINE_SEED = 3


class Measurement(np.ndarray):
    """A ``map2communicate`` array (0.5 outside the footprint, measurement inside) that remembers which footprint,
    altitude level and observation codes it was built from, so it can go back to the device without re-deriving them."""

    def __new__(cls, array, rect=None, alt_index=None, codes=None):
        obj = np.asarray(array).view(cls)
        obj.rect, obj.alt_index, obj.codes = rect, alt_index, codes
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.rect = getattr(obj, "rect", None)
        self.alt_index = getattr(obj, "alt_index", None)
        self.codes = getattr(obj, "codes", None)


class EpisodeEngine:
    def __init__(self, params: Dict, episode: int, device: str = "cuda:0", philox_seed: int = _ENGINE_SEED):
        # maps of this engine can be replaced from outside (set_local / set_global): the K6 area sums are rebuilt on demand
        # (row-major maps: set_local / set_global / the Mapping objects hand them over as the reference's [gx, gy] arrays)
        self.env = VecEnv(params, 1, device=device, philox_seed=philox_seed, track_area=False, map_layout="rows")
        self.d = self.env.d
        self.episode = int(episode)
        env = self.env
        env.episode.fill_(self.episode)
        # truth, start cells, per-episode comm range, prior maps, workspace -- but no sensing yet (agents do that)
        env.ctx.call("ippm_reset_episode", env._p(env.episode), env._p(env.pos), env._p(env.truth), env._p(env.local),
                     env._p(env.glob), env._p(env.split_pct), env._p(env.comm_range), env._p(env.ws), env._p(env.sums), None, 1,
                     env.stream)
        self.start_positions = env.pos[0].cpu().numpy().astype(np.int64)
        self.stage = [0] * self.d.n_agents  # per-agent sensing counter = Philox stage
        # cells any sensing of this episode has covered (evaluation metrics: an unobserved cell is exactly 0.5 in the reference
        # too and never counts as "> 0.5", whereas a cell whose observations cancel is classified by rounding noise)
        self.observed = torch.zeros(self.d.grid_x, self.d.grid_y, dtype=torch.bool, device=env.device)
        raw_sense = env.sense

        def sense(stage, flips=None, agent=-1, close_step=False):
            raw_sense(stage, flips, agent, close_step)
            rects = env.rect[0].cpu().numpy()
            self._rects = rects          # (measurement_views of the agents just sensed reads them here: one copy, not two)
            for i in (range(self.d.n_agents) if agent < 0 else [agent]):
                yu, yd, xl, xr = (int(v) for v in rects[i])
                self.observed[xl:xr, yu:yd] = True

        env.sense = sense

    # ---- maps -------------------------------------------------------------------------------------------
    def _to_logodds(self, prob: np.ndarray) -> torch.Tensor:
        src = torch.from_numpy(np.ascontiguousarray(prob, dtype=np.float32)).to(self.env.device)
        dst = torch.empty_like(src)
        self.env.ctx.call("ippm_prob_to_logodds", _ffi.ptr(src), _ffi.ptr(dst), src.numel(), self.env.stream)
        return dst

    def _to_prob(self, logodds: torch.Tensor) -> np.ndarray:
        return self.env._to_prob(logodds.contiguous()).cpu().numpy()

    def get_local(self, i: int) -> np.ndarray:
        return self._to_prob(self.env.local[0, i])

    def set_local(self, i: int, prob: np.ndarray):
        self.env.local[0, i].copy_(self._to_logodds(prob))
        # an externally supplied map may hold out-of-range values anywhere: make the next fusion clip it all
        ws = self.env.ws[0, i]
        ws[0] = 1
        ws[1:5] = torch.tensor([0, self.d.grid_y, 0, self.d.grid_x], dtype=torch.int32, device=self.env.device)

    def get_global(self) -> np.ndarray:
        return self._to_prob(self.env.glob[0])

    def set_global(self, prob: np.ndarray):
        env = self.env
        env.glob[0].copy_(self._to_logodds(prob))
        ws = env.ws[0, self.d.n_agents]
        ws[0] = 1
        ws[1:5] = torch.tensor([0, self.d.grid_y, 0, self.d.grid_x], dtype=torch.int32, device=env.device)
        out = torch.zeros(1, dtype=torch.float64, device=env.device)
        env.ctx.call("ippm_weighted_entropy", env._p(env.glob), None, 1, _ffi.ptr(out), 1, env.stream)
        env.sums[0, 2] = out[0]

    # ---- measurements -----------------------------------------------------------------------------------
    def measurement_views(self, i: int, position=None):
        """(map2communicate [gx,gy] float32 Measurement, footprint_img [2r,2r] float64, clipped rect, cell_update view).
        ``position``: the agent's position if the caller holds it on the host (else read back from the device)."""
        d, env = self.d, self.env
        rects = getattr(self, "_rects", None)
        self._rects = None               # (valid for the sensing that has just run only)
        yu, yd, xl, xr = (int(v) for v in (rects[i] if rects is not None else env.rect[0, i].cpu()))
        pos = np.asarray(position, dtype=np.int64) if position is not None else env.pos[0, i].cpu().numpy()
        k = min(max((int(pos[2]) - d.min_altitude) // d.spacing, 0), d.space_z - 1)
        codes = d.unpack_tile([yu, yd, xl, xr], env.code[0, i].cpu().numpy())
        meas = np.where(codes > 0, d.meas_value[k, 1], d.meas_value[k, 0]).astype(np.float32)
        m2c = np.full((d.grid_x, d.grid_y), 0.5, dtype=np.float32)
        m2c[xl:xr, yu:yd] = meas
        full, _ = d.footprint(pos)
        img = np.ones((full[1] - full[0], full[3] - full[2])) * 0.5
        hx, wy = xr - xl, yd - yu
        xo = (full[3] - full[2]) - hx if xl > full[2] else 0
        yo = (full[1] - full[0]) - wy if yu > full[0] else 0
        img[xo: xo + hx, yo: yo + wy] = meas
        return Measurement(m2c, rect=[yu, yd, xl, xr], alt_index=k, codes=codes), img, [yu, yd, xl, xr]

    def load_measurement(self, slot: int, m2c: np.ndarray):
        """Puts a map2communicate array back into measurement slot ``slot`` (rect, altitude level, codes)."""
        d, env = self.d, self.env
        rect, k, codes = getattr(m2c, "rect", None), getattr(m2c, "alt_index", None), getattr(m2c, "codes", None)
        if rect is None or codes is None or k is None:  # a plain array: recover footprint, level and codes from its values
            arr = np.asarray(m2c)
            hit = np.abs(arr - 0.5) > 0.01
            if not hit.any():
                rect, k, codes = [0, 0, 0, 0], 0, np.zeros((0, 0), dtype=np.uint8)
            else:
                xs, ys = np.where(hit.any(axis=1))[0], np.where(hit.any(axis=0))[0]
                xl, xr, yu, yd = int(xs[0]), int(xs[-1]) + 1, int(ys[0]), int(ys[-1]) + 1
                vals = arr[xl:xr, yu:yd]
                hi = float(vals.max()) if (vals > 0.5).any() else 1.0 - float(vals.min())
                k = int(np.argmin(np.abs(d.meas_value[:, 1] - hi)))
                rect, codes = [yu, yd, xl, xr], (vals > 0.5).astype(np.uint8)
        yu, yd, xl, xr = rect
        env.code[0, slot].copy_(torch.from_numpy(d.pack_tile(rect, codes)).to(env.device))
        env.rect[0, slot].copy_(torch.tensor(rect, dtype=torch.int32))
        # altitude level of the slot is read from pos[...,2] by the kernels
        env.pos[0, slot, 2] = d.altitudes[k]


_engines: Dict[int, EpisodeEngine] = {}


def scratch_engine(params: Dict) -> EpisodeEngine:
    """A shared engine for stateless calls (Camera, AgentActionSpace, get_global_reward, ...)."""
    key = id(params)
    if key not in _engines:
        _engines[key] = EpisodeEngine(params, 1)
    return _engines[key]


def engine_for_grid(shape) -> Optional[EpisodeEngine]:
    """An already created scratch engine whose grid has this shape (helpers that receive bare maps, e.g. get_wrmse)."""
    for eng in _engines.values():
        if (eng.d.grid_x, eng.d.grid_y) == tuple(int(v) for v in shape):
            return eng
    return None
