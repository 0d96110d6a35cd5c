"""Actor observation [11,11,7] through the K6 feature kernel (reference: actor/transformations.py:14-176)."""
import torch

from .. import _ffi


def get_network_input(local_information, fused_local_map, simulated_map, agent_id, t, params, batch_memory, agent_state_space):
    """Planes: budget, agent id, egocentric positions, weighted entropy of the resized fused local map, weighted entropy
    of the resized footprint image, resized probabilities, resized footprint indicator.  The kernel reads the agent's
    fused map, the published measurements and the received set straight from the episode engine."""
    engine = local_information[agent_id]["engine"]
    env = engine.env
    obs = env.build_features_only(t)
    return obs[0, agent_id].clone()
