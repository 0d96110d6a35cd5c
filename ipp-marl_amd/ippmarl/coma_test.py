"""Deployment of a trained actor with the reference's class name and interface (coma_test.py:29-203):
``COMATest(params, writer, num_episode).execute(test_mode, num_episode) ->
(return, agent_positions, agent_altitudes, entropies, f1s, relative_return)``.

Per step: publish / receive / fuse / actor features (``COMAWrapper.build_observations``), one batched actor forward,
greedy ``argmax(probs * mask)`` per agent with the order-dependent collision mask, move, sense, fuse the new
measurements into the global map, evaluate target entropy and F1.  The two returns are constants in the reference (it
scores ``get_global_reward(current, next)`` after ``current = next.copy()``, coma_test.py:157-174): kept.

The reference unpickles its actor from a fixed absolute path; here ``model_path`` (argument, or
``params["experiment"]["test_model_path"]`` when present) names the file, read with the alias unpickler of
``ippmarl.checkpoint``; without one the wrapper's freshly initialised actor flies (useful for smoke tests only)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ._mission import MissionMetrics, save_mission_numbers  # noqa: F401
from .agent.action_space import AgentActionSpace
from .agent.agent import Agent
from .agent.state_space import AgentStateSpace
from .batch_memory import BatchMemory
from .coma_wrapper import COMAWrapper, ReplayHooks
from .mapping.grid_maps import GridMap
from .mapping.mappings import Mapping
from .sensors import Sensor
from .sensors.models import SensorModel


class COMATest(MissionMetrics):
    def __init__(self, params: Dict, writer, num_episode, model_path: Optional[str] = None):
        self.params = params
        self.num_episode = num_episode
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.x_dim = params["environment"]["x_dim"]
        self.y_dim = params["environment"]["y_dim"]
        self.altitude = params["experiment"]["baselines"]["lawnmower"]["altitude"]
        self.budget = params["experiment"]["constraints"]["budget"]
        self.action_space = AgentActionSpace(params)
        self.coma_wrapper = COMAWrapper(params, writer)
        self.grid_map = GridMap(params)
        self.sensor = Sensor(SensorModel(), self.grid_map)
        self.mapping = Mapping(self.grid_map, self.sensor, params, num_episode)
        self.agent_state_space = AgentStateSpace(params)
        self.map = self.mapping.init_priors()
        self.device = self.coma_wrapper.device
        self.batch_memory = BatchMemory(params, self.coma_wrapper)
        self.model_path = model_path or params["experiment"].get("test_model_path")
        self.net = None          # set to an ActorNetwork to bypass model_path
        self.replay = None       # optional ReplayHooks (parity tests)
        self.f1_bracket = []
        self.greedy_actions = []  # what argmax chose, per (t, agent), also when a replay hook overrides it

    def _load_net(self):
        if self.net is not None:
            return self.net
        if self.model_path:
            from .checkpoint import load_reference_actor
            return load_reference_actor(self.model_path, self.params)
        return self.coma_wrapper.actor_network

    def execute(self, test_mode, num_episode):
        hooks = self.replay or ReplayHooks()
        self.coma_wrapper.replay = hooks
        net = self._load_net().to(self.device)
        net.eval()
        n = self.n_agents
        agents = [Agent(self.coma_wrapper.actor_network, self.params, self.mapping, i, self.agent_state_space) for i in range(n)]
        self.agents = agents
        entropy, f1 = self._metrics()
        entropies, f1s = [entropy], [f1]
        rewards, relative_rewards, agent_positions, agent_altitudes = [], [], [], []
        for t in range(self.budget + 1):
            _, positions, observations = self.coma_wrapper.build_observations(
                self.mapping, agents, num_episode, t, self.params, self.batch_memory, None)
            if t == 0:
                self._fuse_global()            # the start-position measurements
                agent_positions.append([np.asarray(p) for p in positions])
            with torch.no_grad():
                probs, _ = net.forward(torch.stack([o.to(self.device).float() for o in observations]), 0)
            probs = probs.cpu()
            next_positions, altitudes = [], []
            for i, agent in enumerate(agents):
                mask, _ = agent.action_space.get_action_mask(agent.position)
                mask = self.action_space.apply_collision_mask(agent.position, mask, next_positions, self.agent_state_space)
                greedy = int(torch.argmax(probs[i] * torch.tensor(mask)))
                self.greedy_actions.append(greedy)
                forced = hooks.action(i, t)
                agent.position = agent.action_space.action_to_position(agent.position, greedy if forced is None else int(forced))
                agent._sense(hooks.correctness(i, t + 1))
                next_positions.append(agent.position)
                altitudes.append(int(agent.position[2]))
            agent_altitudes.append(altitudes)
            agent_positions.append(next_positions)
            self._fuse_global()                # the measurements just taken
            rewards.append(10 * 0.0 - 0.17)    # zero entropy reduction between a map and its copy
            relative_rewards.append(22 * 0.0 - 0.5)
            entropy, f1 = self._metrics()
            entropies.append(entropy)
            f1s.append(f1)
        return sum(rewards), agent_positions, agent_altitudes, entropies, f1s, sum(relative_rewards)


def main(config_path=None, model_path=None, out_path="coma_test_f1.json"):
    """coma_test.main (coma_test.py:225-304) in "random" start mode: trials, json dump, return / altitude statistics."""
    from .params import load_params
    params = load_params(config_path)
    trials = params["experiment"]["baselines"]["information_gain"]["trials"]
    budget = params["experiment"]["constraints"]["budget"]
    returns, relative_returns, entropies_list, f1_list, altitude_list = [], [], [], [], []
    for trial in range(1, trials + 1):
        ret, _, altitudes, entropies, f1s, rel = COMATest(params, None, trial, model_path).execute("random", trial)
        returns.append(ret)
        relative_returns.append(rel)
        entropies_list.append(entropies)
        f1_list.append(f1s)
        altitude_list += [a for step in altitudes for a in step]
    save_mission_numbers(entropies_list, f1_list, trials, budget, out_path)
    levels = sorted(set(altitude_list))
    counts = [altitude_list.count(v) for v in levels]
    return {"mean_return": float(np.mean(returns)), "max_return": float(np.max(returns)), "min_return": float(np.min(returns)),
            "std_return": float(np.std(returns)), "mean_relative_return": float(np.mean(relative_returns)),
            "altitude_counts": dict(zip(levels, counts)),
            "altitude_ratio": {v: c / sum(counts) for v, c in zip(levels, counts)}}


if __name__ == "__main__":
    print(main())
