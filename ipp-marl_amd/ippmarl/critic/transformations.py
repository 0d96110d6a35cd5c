"""Critic state [11,11,12] through the K6 feature kernel (reference: critic/transformations.py:17-132)."""
import numpy as np
import torch

from .. import _ffi


def get_network_input(t, global_information, accumulated_map_knowledge, batch_memory, agent_id, simulated_map, params):
    """Planes 0-6: the agent's actor observation; 7: allocentric positions; 8/9: weighted entropy / probabilities of the
    resized fused global map; 10: union of the published footprints; 11: the other agents' chosen actions at their
    pre-move cells.  ``global_information`` carries the pre-move positions and published measurements."""
    engine = global_information[agent_id]["engine"]
    env, n = engine.env, engine.d.n_agents
    dev = env.device
    # The reference's loop asks for the agents' states one after the other with the same inputs (coma_wrapper.py:143-160): the
    # kernel builds the whole team's states anyway, so the first call of a sweep keeps them for the agents that follow (a call
    # that does not continue the sweep -- other memory, other step, an agent id that does not ascend -- builds afresh).
    key = (id(batch_memory), id(global_information), t)
    team = getattr(engine, "_critic_states", None)
    if team is None or team[0] != key or agent_id <= team[1]:
        pos_pre = torch.tensor(np.array([np.asarray(global_information[j]["position"]) for j in range(n)], dtype=np.int32)[None]).to(dev)
        rect_pre = torch.tensor(np.array([global_information[j]["map2communicate"].rect for j in range(n)], dtype=np.int32)[None]).to(dev)
        actions = torch.stack([torch.as_tensor(batch_memory.get(-1, j, "action")).reshape(()).to(dev) for j in range(n)]).to(torch.int32).view(1, n)
        obs = torch.stack([batch_memory.get(-1, j, "observation").to(dev).float() for j in range(n)])[None].contiguous()
        state = torch.empty(1, n, 11, 11, 12, dtype=torch.float32, device=dev)
        env.rebuild_area(local=False, glob=True)   # the engine's maps can be replaced from outside: sums from scratch
        env.ctx.call("ippm_critic_features", _ffi.ptr(env.area), _ffi.ptr(rect_pre), _ffi.ptr(pos_pre), _ffi.ptr(actions), _ffi.ptr(obs),
                     _ffi.ptr(state), 1, env.stream)
        team = [key, agent_id, state]
        engine._critic_states = team
    team[1] = agent_id
    out = team[2][0, agent_id].clone()
    batch_memory.insert(-1, agent_id, state=out)
    return out
