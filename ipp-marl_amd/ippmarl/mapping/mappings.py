"""Mapping: ground truth of one episode + sense/update/fuse on the GPU (reference: mapping/mappings.py:19-132).

NumPy arrays in, NumPy arrays out, same signatures and mutation conventions as the reference: ``update_grid_map``
mutates ``map_state`` in place and returns the 5-tuple; ``fuse_map`` never mutates its inputs."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .. import _ffi
from .._engine import EpisodeEngine, Measurement
from .grid_maps import GridMap


class _SimulationView:
    """``mapping.simulation`` of the reference, reduced to what callers read."""

    def __init__(self, simulated_map):
        self.simulated_map = simulated_map


class Mapping:
    def __init__(self, grid_map: GridMap, sensor, params: Dict, episode: int, device: str = "cuda:0"):
        self.params = params
        self.grid_map = grid_map
        self.sensor = sensor
        self.prior = params["mapping"]["prior"]
        self.engine = EpisodeEngine(params, episode, device=device)
        self.simulated_map = self.engine.env.truth_map[0].numpy().astype(np.float64)
        self.simulation = _SimulationView(self.simulated_map)
        self._scratch_calls = 0

    def init_priors(self) -> np.ndarray:
        return np.full((int(self.grid_map.x_dim), int(self.grid_map.y_dim)), self.prior, dtype="float32")

    # ------------------------------------------------------------------------------------------------
    def update_grid_map(self, position, map_state: np.ndarray, t, mode, agent_id: Optional[int] = None,
                        correctness: Optional[np.ndarray] = None):
        """Sense at ``position`` and Bayes-update ``map_state`` in place (K3).  ``agent_id`` selects the engine slot
        and the Philox stream (default slot 0); ``correctness`` (1 = observed correctly) injects the sensor noise."""
        eng, env = self.engine, self.engine.env
        i = 0 if agent_id is None else int(agent_id)
        env.pos[0, i].copy_(torch.as_tensor(np.asarray(position, dtype=np.int32)))
        eng.set_local(i, map_state)
        flips = None
        if correctness is not None:
            _, fc = eng.d.footprint(position)
            flips = self._pack_one(i, fc, 1 - np.asarray(correctness))
        stage = eng.stage[i]
        eng.stage[i] += 1
        env.sense(stage=stage, flips=flips, agent=i)
        new_map = eng.get_local(i)
        m2c, img, fc = eng.measurement_views(i)
        map_state[...] = new_map.astype(map_state.dtype)
        cell_update = map_state[fc[2]:fc[3], fc[0]:fc[1]]
        return map_state, cell_update, fc, m2c, img

    def _pack_one(self, i, fc, flips_tile):
        env, d = self.engine.env, self.engine.d
        out = np.zeros((1, d.n_agents, d.tile_bytes), dtype=np.uint8)
        out[0, i] = d.pack_tile(fc, flips_tile)
        return torch.from_numpy(out).to(env.device)

    # ------------------------------------------------------------------------------------------------
    def fuse_map(self, own_map_state: np.ndarray, other_map_states, agent_id, fusion_mode: str) -> np.ndarray:
        """Stand-alone fusion of measurements into a copy of ``own_map_state`` (K4/K5 kernels, stateless).
        ``other_map_states``: dict {agent: info with "map2communicate"} or a list of map2communicate arrays."""
        eng, env, d = self.engine, self.engine.env, self.engine.d
        if isinstance(other_map_states, dict):
            items = [(k, v["map2communicate"]) for k, v in other_map_states.items() if not (fusion_mode == "local" and k == agent_id)]
        else:
            items = [(None, m) for m in other_map_states]
        # save the engine state this call borrows
        keep = {k: getattr(env, k).clone() for k in ("local", "code", "rect", "pos", "comm", "ws")}
        try:
            cur = torch.from_numpy(np.ascontiguousarray(own_map_state, dtype=np.float32)).to(env.device)
            lo = torch.empty_like(cur)
            env.ctx.call("ippm_prob_to_logodds", _ffi.ptr(cur), _ffi.ptr(lo), cur.numel(), env.stream)
            n = d.n_agents
            if n < 2:
                raise _ffi.IppmError("stand-alone fuse_map needs n_agents >= 2 (one slot for the map, the others for measurements)")
            for base in range(0, len(items), n - 1):
                chunk = items[base: base + n - 1]
                env.ctx.call("ippm_clamp_logodds", _ffi.ptr(lo), lo.numel(), env.stream)  # the op's full-grid input clip
                env.local[0, 0].copy_(lo)
                env.ws[0].zero_()
                env.comm[0].zero_()
                env.comm[0, 0, 0] = 1
                for s, (_, m2c) in enumerate(chunk, start=1):
                    eng.load_measurement(s, m2c)
                    env.comm[0, 0, s] = 1
                env.fuse_local(agent=0)
                lo = env.local[0, 0].clone()
            out = torch.empty_like(lo)
            env.ctx.call("ippm_logodds_to_prob", _ffi.ptr(lo), _ffi.ptr(out), lo.numel(), env.stream)
            return out.cpu().numpy()
        finally:
            for k, v in keep.items():
                getattr(env, k).copy_(v)

    def update_cells(self, map_section, measurement, mode):
        """Bayes update of a free-standing section with a measurement array (reference: mappings.py:106-124)."""
        d = self.engine.d
        sec = np.array(map_section, dtype=np.float32, copy=True)
        full = np.full((d.grid_x, d.grid_y), 0.5, dtype=np.float32)
        h, w = sec.shape
        full[:h, :w] = sec
        meas = np.full((d.grid_x, d.grid_y), 0.5, dtype=np.float32)
        meas[:h, :w] = measurement
        # the reference clips its input in place
        np.clip(map_section, 0.0001, 0.9999, out=map_section)
        return self.fuse_map(full, [Measurement(meas)], None, "global")[:h, :w]
