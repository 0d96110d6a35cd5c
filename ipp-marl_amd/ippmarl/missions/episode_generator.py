"""One episode = budget+1 x (build_observations, steps) (reference: missions/episode_generator.py:17-102)."""
from typing import Dict, List

from ..agent.agent import Agent
from ..mapping.mappings import Mapping


class EpisodeGenerator:
    def __init__(self, params: Dict, writer, grid_map, sensor):
        self.params = params
        self.writer = writer
        self.grid_map = grid_map
        self.sensor = sensor
        self.budget = params["experiment"]["constraints"]["budget"]
        self.n_agents = params["experiment"]["missions"]["n_agents"]

    def execute(self, num_episode: int, batch_memory, coma_wrapper, mode):
        mapping = Mapping(self.grid_map, self.sensor, self.params, num_episode)
        agents = self.init_agents(mapping, coma_wrapper)
        self.agents = agents
        episode_return = absolute_return = 0
        episode_rewards, agent_positions, agent_actions, agent_altitudes = [], [], [], []
        current_global_map = agents[0].local_map.copy()
        eps = None
        for t in range(self.budget + 1):
            global_information, positions, observations = coma_wrapper.build_observations(
                mapping, agents, num_episode, t, self.params, batch_memory, mode)
            (batch_memory, relative_reward, absolute_reward, done, new_positions, eps, actions, altitudes,
             current_global_map) = coma_wrapper.steps(mapping, t, agents, current_global_map, num_episode, batch_memory,
                                                      global_information, mapping.simulated_map, self.params, mode)
            agent_actions.append(actions)
            episode_return += relative_reward
            episode_rewards.append(relative_reward)
            absolute_return += absolute_reward
            if t == 0:
                agent_positions.append(positions)
            agent_positions.append(new_positions)
            agent_altitudes.append(altitudes)
        return (episode_return, episode_rewards, absolute_return, mapping.simulated_map, batch_memory, agent_positions, t, eps,
                agent_actions, agent_altitudes)

    def init_agents(self, mapping: Mapping, coma_wrapper) -> List[Agent]:
        return [Agent(coma_wrapper.actor_network, self.params, mapping, agent_id, coma_wrapper.agent_state_space)
                for agent_id in range(self.n_agents)]
