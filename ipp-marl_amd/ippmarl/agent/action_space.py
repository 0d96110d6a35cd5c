"""Action masks, collision masks and moves through the K1 mask kernel (reference: agent/action_space.py:9-589)."""
from typing import Dict

import numpy as np
import torch

from .. import _ffi
from .._engine import scratch_engine


class AgentActionSpace:
    def __init__(self, params: Dict):
        self.params = params
        con = params["experiment"]["constraints"]
        self.spacing = con["spacing"]
        self.min_altitude, self.max_altitude = con["min_altitude"], con["max_altitude"]
        self.space_x_dim = self.space_y_dim = 3
        self.space_z_dim = (self.max_altitude - self.min_altitude) // self.spacing + 1
        self.num_actions = con["num_actions"]
        self.environment_x_dim = params["environment"]["x_dim"]
        self.environment_y_dim = params["environment"]["y_dim"]
        self.space_dim = np.array([self.space_x_dim, self.space_y_dim, self.space_z_dim])

    def _query(self, position, mask_in=None, others=()):
        env = scratch_engine(self.params).env
        dev, A = env.device, self.num_actions
        pos = torch.as_tensor(np.asarray(position, dtype=np.int32).reshape(1, 3)).to(dev)
        k = max(len(others), 1)
        oth = torch.zeros(1, k, 3, dtype=torch.int32)
        for j, o in enumerate(others):
            oth[0, j] = torch.as_tensor(np.asarray(o, dtype=np.int32))
        oth = oth.to(dev)
        n_o = torch.tensor([len(others)], dtype=torch.int32, device=dev)
        m_in = None if mask_in is None else torch.as_tensor((np.asarray(mask_in) != 0).astype(np.uint8).reshape(1, A)).to(dev)
        m_out = torch.zeros(1, A, dtype=torch.uint8, device=dev)
        nxt = torch.zeros(1, A, 3, dtype=torch.int32, device=dev)
        env.ctx.call("ippm_action_mask", _ffi.ptr(pos), _ffi.ptr(oth), _ffi.ptr(n_o), k, _ffi.ptr(m_in), _ffi.ptr(m_out),
                     _ffi.ptr(nxt), 1, env.stream)
        return m_out[0].cpu().numpy().astype(np.float64), nxt[0].cpu().numpy().astype(np.int64)

    def _tables(self):
        """Boundary masks and next positions of EVERY lattice position, from one batched launch of the mask kernel (kept with the
        params' scratch engine): get_action_mask / action_to_position then read the kernel's answers from the table instead of
        launching it and copying two small arrays back for every agent and step."""
        eng = scratch_engine(self.params)
        tab = getattr(eng, "_action_tables", None)
        if tab is None:
            env, d, A = eng.env, eng.d, self.num_actions
            cells = [(ix * d.spacing, iy * d.spacing, d.min_altitude + iz * d.spacing)
                     for ix in range(d.space_x) for iy in range(d.space_y) for iz in range(d.space_z)]
            pos = torch.tensor(cells, dtype=torch.int32).to(env.device)
            B = len(cells)
            oth = torch.zeros(B, 1, 3, dtype=torch.int32, device=env.device)
            n_o = torch.zeros(B, dtype=torch.int32, device=env.device)
            m_out = torch.zeros(B, A, dtype=torch.uint8, device=env.device)
            nxt = torch.zeros(B, A, 3, dtype=torch.int32, device=env.device)
            env.ctx.call("ippm_action_mask", _ffi.ptr(pos), _ffi.ptr(oth), _ffi.ptr(n_o), 1, None, _ffi.ptr(m_out), _ffi.ptr(nxt), B, env.stream)
            masks, nexts = m_out.cpu().numpy().astype(np.float64), nxt.cpu().numpy().astype(np.int64)
            tab = eng._action_tables = {c: (masks[i], nexts[i]) for i, c in enumerate(cells)}
        return tab

    def _lookup(self, position):
        key = tuple(int(v) for v in np.asarray(position).reshape(-1)[:3])
        hit = self._tables().get(key)
        if hit is None:      # off the lattice: ask the kernel
            return self._query(position)
        return hit[0].copy(), hit[1].copy()

    def get_action_mask(self, position):
        """-> (mask_flatten float64 [A], mask in the reference's grid shape for 9/27 actions)."""
        flat, _ = self._lookup(position)
        shaped = flat
        if self.num_actions == 9:
            shaped = flat.reshape(3, 3)
        elif self.num_actions == 27:
            shaped = flat.reshape(3, 3, 3)
        return flat, shaped

    def action_to_position(self, position: np.array, action_index: int):
        _, nxt = self._lookup(position)
        return nxt[int(action_index)]

    def apply_collision_mask(self, position, mask, next_other_positions, agent_state_space):
        """Zeroes the actions leading to cells taken by already-moved agents; mutates and returns ``mask``."""
        out, _ = self._query(position, mask_in=mask, others=list(next_other_positions))
        mask[...] = out
        return mask
