"""Actor / critic convnets (PyTorch-ROCm; the only place MFMA is used on this path).

Same architecture, layer names and initialisation order as the reference (actor/network.py:10-96,
critic/network.py:12-47) so that a reference ``state_dict`` / ``best_model.pth`` loads unchanged: conv 5x5 -> 4x4 -> 4x4
(11 -> 7 -> 4 -> 1), fc1, (fc2: present but unused, exactly like the reference), fc3.  Inputs are channels-last
[B,11,11,C] as produced by the feature kernels.  The reference's global ``torch.autograd.set_detect_anomaly(True)``
(critic/network.py:9) is deliberately not reproduced.
"""
from __future__ import annotations

import os
from typing import Dict

import torch
from torch import nn

# MIOpen's exhaustive solver search (needed: its fast mode picks kernels that make a COMA update 6x slower) also times the
# naive reference convolutions, which take 0.3-0.5 s per run on the 12k-sample minibatches: 40 s of warm-up per process
# for nothing.  Leave them out of the search unless the user says otherwise.
for _v in ("FWD", "BWD", "WRW"):
    os.environ.setdefault(f"MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_{_v}", "0")


def epsilon_schedule(params: Dict, num_episode: int) -> float:
    """Linear anneal eps_max -> eps_min over eps_anneal_phase episodes (actor/network.py:53-58)."""
    m = params["experiment"]["missions"]
    if num_episode > m["eps_anneal_phase"]:
        return m["eps_min"]
    return m["eps_max"] - num_episode / m["eps_anneal_phase"] * (m["eps_max"] - m["eps_min"])


class _ConvTrunk(nn.Module):
    def __init__(self, in_planes: int, n_actions: int):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, 256, (5, 5))
        self.conv2 = nn.Conv2d(256, 256, (4, 4))
        self.conv3 = nn.Conv2d(256, 256, (4, 4))
        self.activation = nn.ReLU()
        self.flatten = nn.Flatten()
        self.fc1 = nn.Linear(256, 256)
        self.fc2 = nn.Linear(256, 256)  # never used in forward (reference: commented out) -> never gets a gradient
        self.fc3 = nn.Linear(256, n_actions)

    def trunk(self, x: torch.Tensor):
        if x.dim() == 3:
            x = x.unsqueeze(0)
        x = x.permute(0, 3, 1, 2)  # NHWC storage -> logical NCHW (channels_last strides, no copy)
        h = self.activation(self.conv1(x))
        h = self.activation(self.conv2(h))
        h = self.activation(self.conv3(h))
        h = self.flatten(h)
        return self.fc3(self.activation(self.fc1(h))), h


class ActorNetwork(_ConvTrunk):
    def __init__(self, params: Dict):
        self.params = params
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.n_actions = params["experiment"]["constraints"]["num_actions"]
        super().__init__(7, self.n_actions)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, input_state: torch.Tensor, eps: float):
        """-> ((1-eps) * softmax + eps / n_actions, hidden) (actor/network.py:70-88)."""
        logits, h = self.trunk(input_state)
        probs = self.softmax(logits)
        return (1 - eps) * probs + eps / self.n_actions, h

    def eps(self, num_episode: int) -> float:
        return epsilon_schedule(self.params, num_episode)

    def get_action_index(self, batch_memory, action_mask_1d, agent_id, t, num_episode: int, mode: str):
        """Single-agent action choice of the drop-in Agent.step (actor/network.py:41-68,90-96): masked eps-softmax,
        torch.multinomial in training, argmax in evaluation."""
        device = next(self.parameters()).device
        obs = batch_memory.get(-1, agent_id, "observation").unsqueeze(0).to(device).float()
        mask = torch.as_tensor(action_mask_1d).to(device)
        eps = epsilon_schedule(self.params, num_episode)
        with torch.no_grad():
            probs, _ = self.forward(obs, eps)
        probs = probs.squeeze() * mask
        chosen = torch.argmax(probs) if mode == "eval" else torch.multinomial(probs, 1, replacement=True)
        return probs, chosen, mask, eps


class CriticNetwork(_ConvTrunk):
    def __init__(self, params: Dict):
        self.params = params
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.n_actions = params["experiment"]["constraints"]["num_actions"]
        super().__init__(12, self.n_actions)

    def forward(self, input_state: torch.Tensor):
        """-> (Q [B,A], log_softmax over dim 0 (a metric the reference logs; critic/network.py:43-47))."""
        q, _ = self.trunk(input_state)
        q = q.squeeze()
        with torch.no_grad():
            log_probs = torch.log_softmax(q, dim=0)
        return q, log_probs
