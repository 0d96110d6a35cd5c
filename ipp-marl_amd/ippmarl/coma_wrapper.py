"""COMAWrapper with the reference's constructor and method signatures (coma_wrapper.py:22-183): one environment,
reference-shaped objects, every arithmetic step on the GPU kernels.  For throughput use ippmarl.trainer.COMATrainer
(thousands of envs per launch); this class is the drop-in seam for code written against the reference."""
from __future__ import annotations

import copy
from typing import Dict, List, Optional

import numpy as np
import torch

from .actor.transformations import get_network_input as get_actor_input
from .agent.agent import Agent
from .agent.communication_log import CommunicationLog
from .agent.state_space import AgentStateSpace
from .critic.transformations import get_network_input as get_critic_input
from .learners import ActorLearner, CriticLearner
from .networks import ActorNetwork, CriticNetwork
from ._engine import scratch_engine


class ReplayHooks:
    """Optional injection of the randomness a recorded episode consumed (parity tests)."""

    def __init__(self, correctness=None, action=None):
        self.correctness = correctness or (lambda agent_id, stage: None)
        self.action = action or (lambda agent_id, t: None)


class COMAWrapper:
    def __init__(self, params: Dict, writer=None, device: str = "cuda:0"):
        self.params = params
        self.mission_type = params["experiment"]["missions"]["type"]
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.budget = params["experiment"]["constraints"]["budget"]
        self.device = torch.device(device)
        self.agent_state_space = AgentStateSpace(params)
        self.actor_network = ActorNetwork(params)
        ctx = scratch_engine(params).env.ctx
        self.actor_learner = ActorLearner(params, self.actor_network, self.device, ctx=ctx)
        self.critic_network = CriticNetwork(params)
        self.target_critic_network = copy.deepcopy(self.critic_network)
        self.critic_learner = CriticLearner(params, self.critic_network, self.device)
        self.replay: Optional[ReplayHooks] = None

    def build_observations(self, mapping, agents: List[Agent], num_episode, t, params, batch_memory, mode):
        communication_log = CommunicationLog(self.params, num_episode, engine=mapping.engine)
        hooks = self.replay or ReplayHooks()
        positions, global_information = [], {}
        for agent_id in range(self.n_agents):
            global_information, _, position = agents[agent_id].communicate(t, num_episode, communication_log, mode,
                                                                           correctness=hooks.correctness(agent_id, 0) if t == 0 else None)
            positions.append(position)
        observations = []
        for agent_id in range(self.n_agents):
            local_information, fused_local_map = agents[agent_id].receive_messages(communication_log, agent_id, t)
            observation = get_actor_input(local_information, fused_local_map, mapping.simulated_map, agent_id, t, params,
                                          batch_memory, self.agent_state_space)
            batch_memory.add(agent_id, observation=observation)
            observations.append(observation)
        return global_information, positions, observations

    def steps(self, mapping, t: int, agents: List[Agent], accumulated_map_knowledge, num_episode, batch_memory, global_information,
              simulated_map, params, mode):
        engine, env = mapping.engine, mapping.engine.env
        hooks = self.replay or ReplayHooks()
        if accumulated_map_knowledge is not getattr(mapping, "_global_token", None):
            engine.set_global(np.asarray(accumulated_map_knowledge))  # a map this engine did not produce itself
        # global fusion of the published measurements + information-gain reward (K5); the reference computes the same
        # fusion twice (coma_wrapper.py:93 and :145)
        env.ctx.call("ippm_fuse_global_reward", env._p(env.glob), env._p(env.code), env._p(env.rect), env._p(env.pos),
                     env._p(env.ws), env._p(env.sums), env._p(env.reward), 1, env.stream)
        next_positions, actions, footprints, altitudes = [], [], [], []
        eps = None
        for agent_id in range(self.n_agents):
            _, next_position, eps, action, footprint_idx, _ = agents[agent_id].step(
                agent_id, t, num_episode, batch_memory, mode, next_positions,
                correctness=hooks.correctness(agent_id, t + 1), action=hooks.action(agent_id, t))
            next_positions.append(next_position)
            actions.append(int(action))
            footprints.append(footprint_idx)
            altitudes.append(next_position[2])
        for agent_id in range(self.n_agents):
            get_critic_input(t, global_information, None, batch_memory, agent_id, simulated_map, params)
        next_global_map = engine.get_global()
        mapping._global_token = next_global_map
        relative_reward, absolute_reward = (float(v) for v in env.reward[0].cpu())
        done = t == self.budget
        for agent_id in range(self.n_agents):
            batch_memory.insert(-1, agent_id, reward=relative_reward, done=done)
        return (batch_memory, relative_reward, absolute_reward, done, next_positions, eps, actions, altitudes, next_global_map)
