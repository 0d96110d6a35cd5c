"""Range-limited message exchange (reference: agent/communication_log.py:11-65); the in-range test runs in the comm
kernel, the per-link failure draw uses np.random.random_sample() like the reference (one draw per ordered pair)."""
from typing import Dict

import numpy as np
import torch

from .. import _ffi
from .._engine import scratch_engine


class CommunicationLog:
    def __init__(self, params: Dict, num_episode: int, engine=None):
        self.params = params
        uav = params["experiment"]["uav"]
        self.communication_range = uav["communication_range"]
        self.fix_range = uav["fix_range"]
        self.failure_rate = uav["failure_rate"]
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.global_log = dict()
        self._engine = engine
        if not self.fix_range:  # np.random.seed(episode); randint(4) -> {0,15,25,100}: same MT19937 routine as the reset kernel
            split, _ = _ffi.host_truth_params(int(num_episode))
            self.communication_range = [0, 15, 25, 100][split]

    def store_agent_message(self, message: Dict, agent_id: int):
        self.global_log[agent_id] = message
        return self.global_log

    def received_row(self, agent_id: int):
        env = (self._engine or scratch_engine(self.params)).env
        n = self.n_agents
        ids = sorted(self.global_log.keys())
        pos = torch.zeros(1, n, 3, dtype=torch.int32)
        for j in ids:
            pos[0, j] = torch.as_tensor(np.asarray(self.global_log[j]["position"], dtype=np.int32))
        pos = pos.to(env.device)
        draws = torch.tensor([[[np.random.random_sample() if (i == agent_id and j in self.global_log) else 1.0
                                for j in range(n)] for i in range(n)]], dtype=torch.float64, device=env.device)
        rng = torch.tensor([float(self.communication_range)], dtype=torch.float32, device=env.device)
        comm = torch.zeros(1, n, n, dtype=torch.uint8, device=env.device)
        env.ctx.call("ippm_comm_matrix", None, _ffi.ptr(pos), _ffi.ptr(rng), _ffi.ptr(draws), _ffi.ptr(comm), 0, 1, env.stream)
        row = comm[0, agent_id].cpu().numpy()
        return [j for j in ids if row[j]]

    def get_messages(self, agent_id: int):
        return {j: self.global_log[j] for j in self.received_row(agent_id)}

    def get_global_positions(self):
        return [[self.global_log[a]["position"]] for a in self.global_log]
