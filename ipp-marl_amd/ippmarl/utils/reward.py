"""Information-gain reward of two maps on the GPU (reference: utils/reward.py:11-82, utils/state.py:53-121)."""
import numpy as np
import torch

from .. import _ffi
from .._engine import scratch_engine


def get_utility_reward(state: np.array, state_: np.array, simulated_map, agent_state_space, params=None):
    """-> (absolute, relative) = (S1 / cells, S1 / S2) with S1 = sum w(a)(H(b)-H(a)), S2 = sum w(a) H(b)."""
    params = params if params is not None else agent_state_space.params
    env = scratch_engine(params).env
    dev = env.device

    def up(m):
        src = torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to(dev)
        dst = torch.empty_like(src)
        env.ctx.call("ippm_prob_to_logodds", _ffi.ptr(src), _ffi.ptr(dst), src.numel(), env.stream)
        return dst

    before, after = up(state), up(state_)
    sums = torch.zeros(1, 2, dtype=torch.float64, device=dev)
    env.ctx.call("ippm_reward_from_maps", _ffi.ptr(before), _ffi.ptr(after), _ffi.ptr(sums), None, 1, env.stream)
    s1, s2 = (float(v) for v in sums[0].cpu())
    absolute = s1 / before.numel()
    return absolute, s1 / s2


def get_global_reward(last_map, next_map, mission_type, footprints, simulated_map, agent_state_space, actions, agent_id, t,
                      budget):
    """-> (done=False, 22 * relative - 0.5, 10 * absolute - 0.17)."""
    absolute, relative = get_utility_reward(last_map, next_map, simulated_map, agent_state_space)
    return False, 22 * relative - 0.5, 10 * absolute - 0.17
