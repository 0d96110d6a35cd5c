"""UAV agent: publish, receive+fuse (K4), act+move (K1 mask kernel + actor), sense (K3)
(reference: agent/agent.py:13-117)."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .action_space import AgentActionSpace
from .communication_log import CommunicationLog
from .state_space import AgentStateSpace


class LazyMap:
    """A map that lives on the device and is downloaded (as probabilities) only when somebody looks at it."""

    def __init__(self, fetch):
        self._fetch = fetch

    def __array__(self, dtype=None, copy=None):
        a = self._fetch()
        return a if dtype is None else a.astype(dtype)

    def copy(self):
        return self._fetch().copy()


class Agent:
    def __init__(self, actor_network, params: Dict, mapping, agent_id: int, agent_state_space: AgentStateSpace):
        self.params = params
        self.agent_id = agent_id
        self.mission_type = params["experiment"]["missions"]["type"]
        self.n_actions = params["experiment"]["constraints"]["num_actions"]
        self.x_dim, self.y_dim = params["environment"]["x_dim"], params["environment"]["y_dim"]
        self.mapping = mapping
        self.engine = mapping.engine
        self.agent_state_space = agent_state_space
        self.action_space = AgentActionSpace(params)
        self.actor_network = actor_network
        self.agent_info = dict()
        self.position = None
        self.map_footprint = None
        self.map2communicate = None
        self.footprint_img = None
        self.local_map = mapping.init_priors()   # goes through the setter: uploads the prior

    # the local map lives in engine slot ``agent_id``
    @property
    def local_map(self):
        return self.engine.get_local(self.agent_id)

    @local_map.setter
    def local_map(self, value):
        if not isinstance(value, LazyMap):
            self.engine.set_local(self.agent_id, np.asarray(value))

    def _sense(self, correctness=None):
        """K3 on the agent's own device-resident map at its current position."""
        eng, env, i = self.engine, self.engine.env, self.agent_id
        env.pos[0, i].copy_(torch.as_tensor(np.asarray(self.position, dtype=np.int32)))
        flips = None
        if correctness is not None:
            _, fc = eng.d.footprint(self.position)
            flips = self.mapping._pack_one(i, fc, 1 - np.asarray(correctness))
        stage = eng.stage[i]
        eng.stage[i] += 1
        env.sense(stage=stage, flips=flips, agent=i)
        self.map2communicate, self.footprint_img, fc = eng.measurement_views(i, self.position)
        self.map_footprint = LazyMap(lambda: self.engine.get_local(i)[fc[2]:fc[3], fc[0]:fc[1]])
        return fc

    def communicate(self, t, num_episode, communication_log: CommunicationLog, mode, correctness=None):
        if t == 0:
            self.position = self.agent_state_space.get_random_agent_state(self.agent_id, num_episode)
            self._sense(correctness)
        agent_info = {"local_map": LazyMap(lambda: self.engine.get_local(self.agent_id)), "position": self.position,
                      "map_footprint": self.map_footprint, "map2communicate": self.map2communicate,
                      "footprint_img": self.footprint_img, "engine": self.engine}
        global_log = communication_log.store_agent_message(agent_info, self.agent_id)
        return global_log, agent_info["local_map"], self.position

    def receive_messages(self, communication_log, agent_id, t):
        received = communication_log.get_messages(self.agent_id)
        env, n = self.engine.env, self.engine.d.n_agents
        row = torch.zeros(n, dtype=torch.uint8)
        for j in received:
            row[j] = 1
        env.comm[0, self.agent_id].copy_(row)
        if len(received) > 0:
            env.fuse_local(agent=self.agent_id)
        return received, LazyMap(lambda: self.engine.get_local(self.agent_id))

    def step(self, agent_id, t, num_episode, batch_memory, mode, next_other_positions, correctness=None, action=None):
        mask_1d, _ = self.action_space.get_action_mask(self.position)
        mask_1d = self.action_space.apply_collision_mask(self.position, mask_1d, next_other_positions, self.agent_state_space)
        probs, chosen, mask, eps = self.actor_network.get_action_index(batch_memory, mask_1d, self.agent_id, t, num_episode, mode)
        if action is not None:  # replay hook: the caller dictates the action (parity tests)
            chosen = torch.as_tensor(action)
        self.position = self.action_space.action_to_position(self.position, int(chosen))
        if not self.is_in_map(self.position):
            print("OUT OF MAP")
        footprint_idx = self._sense(correctness)
        batch_memory.insert(-1, agent_id, action=chosen, mask=mask)
        return (LazyMap(lambda: self.engine.get_local(self.agent_id)), self.position, eps, chosen, footprint_idx,
                self.map2communicate)

    def is_in_map(self, position):
        d = self.engine.d
        return bool(0 <= position[0] <= self.x_dim and 0 <= position[1] <= self.y_dim
                    and d.min_altitude <= position[2] <= d.max_altitude)
