"""Greedy expected-information-gain planner with the reference's class name and interface (IG_baseline.py:32-325):
``IG_baseline(params, writer, num_episode).execute() -> (relative_return, absolute_return, altitudes, entropies, f1s)``.

Candidate evaluation (K9), selection (K10), masks, sensing, fusion and the evaluation metrics run on the GPU; the
class only sequences them the way the reference's ``execute`` does.  (the two returns are constants in the
reference because it compares a map with itself; the informative outputs are the per-step target entropy and F1.)"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import _ffi
from .agent.action_space import AgentActionSpace
from .agent.agent import Agent
from .agent.state_space import AgentStateSpace
from .batch_memory import BatchMemory
from .coma_wrapper import COMAWrapper, ReplayHooks
from ._mission import MissionMetrics
from .mapping.grid_maps import GridMap
from .mapping.mappings import Mapping
from .sensors import Sensor
from .sensors.cameras import Camera
from .sensors.models.sensor_models import AltitudeSensorModel


class IG_baseline(MissionMetrics):
    def __init__(self, params: Dict, writer, num_episode):
        self.params = params
        self.num_episode = num_episode
        self.budget = params["experiment"]["constraints"]["budget"]
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.class_weighting = params["experiment"]["missions"]["class_weighting"]
        self.communication = params["experiment"]["baselines"]["information_gain"]["communication"]
        self.coma_wrapper = COMAWrapper(params, writer)
        self.grid_map = GridMap(params)
        self.sensor_model = AltitudeSensorModel(params)
        self.sensor = Sensor(self.sensor_model, self.grid_map)
        self.mapping = Mapping(self.grid_map, self.sensor, params, num_episode)
        self.agent_state_space = AgentStateSpace(params)
        self.action_space = AgentActionSpace(params)
        self.batch_memory = BatchMemory(params, self.coma_wrapper)
        self.camera = Camera(params, self.sensor_model, self.grid_map)
        self.writer = writer
        self.replay = None        # optional ReplayHooks(correctness=...) for parity tests
        self.gains_log = []
        self.f1_bracket = []      # per evaluation: F1 with the exactly-cancelled cells counted as free / as occupied

    def get_individual_ig(self, position, action_mask, map_state=None, agent_id: int = 0):
        """Expected information gain per action for the agent in engine slot ``agent_id`` (its device-resident local map;
        ``map_state`` is accepted for signature compatibility and uploaded when given)."""
        eng, env = self.mapping.engine, self.mapping.engine.env
        if map_state is not None and not hasattr(map_state, "_fetch"):
            eng.set_local(agent_id, np.asarray(map_state))
        n, A = self.n_agents, len(action_mask)
        pos = env.pos.clone()
        pos[0, agent_id] = torch.as_tensor(np.asarray(position, dtype=np.int32))
        mask = torch.zeros(1, n, A, dtype=torch.uint8, device=env.device)
        mask[0, agent_id] = torch.as_tensor((np.asarray(action_mask) != 0).astype(np.uint8))
        gains = torch.zeros(1, n, A, dtype=torch.float32, device=env.device)
        env.ctx.call("ippm_ig_candidates", env._p(env.local), _ffi.ptr(pos), _ffi.ptr(mask), _ffi.ptr(gains), 1, env.stream)
        g = gains[0, agent_id].cpu().numpy().astype(np.float64)
        positions = [self.action_space.action_to_position(position, a) if action_mask[a] != 0 else 0 for a in range(A)]
        return positions, [float(v) if action_mask[a] != 0 else 0 for a, v in enumerate(g)]

    def execute(self):
        eng, env = self.mapping.engine, self.mapping.engine.env
        n, A = self.n_agents, self.params["experiment"]["constraints"]["num_actions"]
        dev = env.device
        hooks = self.replay or ReplayHooks()
        self.coma_wrapper.replay = hooks
        agents = [Agent(self.coma_wrapper.actor_network, self.params, self.mapping, i, self.agent_state_space) for i in range(n)]
        self.agents = agents
        entropy, f1 = self._metrics()
        entropies, f1s = [entropy], [f1]
        agent_positions, agent_altitudes, relative_rewards, absolute_rewards = [], [], [], []
        for t in range(self.budget + 1):
            global_information, positions, _ = self.coma_wrapper.build_observations(
                self.mapping, agents, self.num_episode, t, self.params, self.batch_memory, None)
            if t == 0:
                agent_positions.append(positions)
                self._fuse_global()   # the start-position measurements
            # masks: agent i is masked against the CURRENT positions of the agents before it (IG_baseline.py:127-148)
            masks = np.zeros((n, A))
            prior_positions = []
            for i in range(n):
                m = self.action_space.get_action_mask(agents[i].position)[0]
                masks[i] = self.action_space.apply_collision_mask(agents[i].position, m, prior_positions, self.agent_state_space)
                prior_positions.append(agents[i].position)
            mask_t = torch.as_tensor(masks.astype(np.uint8)[None]).to(dev)
            gains = torch.zeros(1, n, A, dtype=torch.float32, device=dev)
            env.ctx.call("ippm_ig_candidates", env._p(env.local), env._p(env.pos), _ffi.ptr(mask_t), _ffi.ptr(gains), 1, env.stream)
            self.gains_log.append(gains[0].cpu().numpy().astype(np.float64))
            actions = torch.zeros(1, n, dtype=torch.int32, device=dev)
            env.ctx.call("ippm_ig_select", env._p(env.pos), _ffi.ptr(mask_t), _ffi.ptr(gains), 1 if self.communication else 0,
                         _ffi.ptr(actions), None, 1, env.stream)
            chosen = actions[0].cpu().numpy()
            next_positions = list(prior_positions)
            altitudes = []
            for i in range(n):
                agents[i].position = self.action_space.action_to_position(agents[i].position, int(chosen[i]))
                agents[i]._sense(hooks.correctness(i, t + 1))
                next_positions.append(agents[i].position)
                altitudes.append(int(agents[i].position[2]))
            agent_positions.append(next_positions)
            agent_altitudes.append(altitudes)
            self._fuse_global()       # the measurements just taken (no one-step lag in this script)
            # the reference evaluates get_global_reward(current, next) with current == next: zero entropy reduction
            relative_rewards.append(22 * 0.0 - 0.5)
            absolute_rewards.append(10 * 0.0 - 0.17)
            entropy, f1 = self._metrics()
            entropies.append(entropy)
            f1s.append(f1)
        return sum(relative_rewards), sum(absolute_rewards), agent_altitudes, entropies, f1s
