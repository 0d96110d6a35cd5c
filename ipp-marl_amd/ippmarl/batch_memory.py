"""Transition memory with the reference's interface (batch_memory.py:12-191); TD(lambda) targets come from one batched
target-critic forward plus the K8 kernel instead of ~2400 batch-1 forwards."""
from typing import Dict

import numpy as np
import torch

from . import _ffi
from ._engine import scratch_engine
from .utils.utils import TransitionCOMA


class BatchMemory:
    def __init__(self, params: Dict, coma_network):
        self.params = params
        self.coma_network = coma_network
        self.batch_size = params["networks"]["batch_size"]
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.budget = params["experiment"]["constraints"]["budget"]
        self.gamma = params["networks"]["gamma"]
        self.lam = params["networks"]["lambda"]
        self.transitions = {a: [] for a in range(self.n_agents)}

    def clear(self):
        self.transitions = {a: [] for a in range(self.n_agents)}

    def add(self, agent_id: int, state=None, observation=None, action=None, mask=None, reward=None, done=None, td_target=None,
            discounted_return=None):
        self.transitions[agent_id].append(TransitionCOMA(state, observation, action, mask, reward, done, td_target,
                                                         discounted_return))

    def insert(self, t: int, agent_id: int, **fields):
        fields = {k: v for k, v in fields.items() if v is not None}
        self.transitions[agent_id][t] = self.transitions[agent_id][t]._replace(**fields)

    def get(self, t: int, agent_id: int, argument: str):
        return getattr(self.transitions[agent_id][t], argument)

    def size(self):
        return len(self.transitions[0]) * self.n_agents

    def build_td_targets(self, target_critic_network):
        n, L = self.n_agents, len(self.transitions[0])
        env = scratch_engine(self.params).env
        dev = env.device
        states = torch.stack([self.transitions[a][t].state for a in range(n) for t in range(L)]).to(dev).float()
        actions = torch.tensor([int(self.transitions[a][t].action) for a in range(n) for t in range(L)], device=dev)
        with torch.no_grad():
            q, _ = target_critic_network.to(dev)(states)
        q_sel = q.view(n * L, -1).gather(1, actions.view(-1, 1)).view(n, L).contiguous().float()
        rew = torch.tensor([[float(self.transitions[a][t].reward) for t in range(L)] for a in range(n)], dtype=torch.float32, device=dev)
        done = torch.tensor([[1 if self.transitions[a][t].done else 0 for t in range(L)] for a in range(n)], dtype=torch.uint8, device=dev)
        td, dr = torch.empty_like(rew), torch.empty_like(rew)
        env.ctx.call("ippm_td_lambda", _ffi.ptr(rew), _ffi.ptr(done), _ffi.ptr(q_sel), _ffi.ptr(td), _ffi.ptr(dr), n, L, env.stream)
        td, dr = td.cpu(), dr.cpu()
        for a in range(n):
            for t in range(L):
                self.insert(t, a, td_target=td[a, t].view(1), discounted_return=dr[a, t].view(1))

    def build_batches(self):
        usable = self.size() - self.size() % self.batch_size
        idx = np.arange(0, usable, dtype=np.int32)
        np.random.shuffle(idx)
        flat = self.concatenated_transitions
        return [[flat[i] for i in idx[s: s + self.batch_size]] for s in range(0, usable, self.batch_size)]

    @property
    def concatenated_transitions(self):
        return [self.transitions[a][t] for t in range(len(self.transitions[0])) for a in range(self.n_agents)]
