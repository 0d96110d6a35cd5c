"""The diagnostics the reference logs to TensorBoard once per training step (missions/coma_mission.py:208-424), computed
from what the learners saw during data pass 0, under the reference's scalar names.

Definitions follow critic/learner.py:58-190 and actor/learner.py:36-200 literally, including the odd ones:
* "Critic/Log probs according to critic": the critic's second output is log_softmax over the BATCH axis (dim 0);
* "Actor/KL divergence policy": rel_entr(pi_old, exp(pi_new)) -- the script exponentiates probabilities -- summed over the
  minibatches of the pass, then averaged over (sample, action);
* "Actor/Hidden state entropy": np.square called with two arguments, i.e. the mean square of the FIRST sample's
  flattened conv features of each minibatch;
* gradient figures: L1 norm of the last minibatch's gradients, averaged over each layer's (weight, bias); fc2 is unused: 0.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

LAYERS = ("conv1", "conv2", "conv3", "fc1", "fc2", "fc3")
LAYER_TAGS = ("Conv1", "Conv2", "Conv3", "FC1", "FC2", "FC3")


def grad_l1_norms(net: torch.nn.Module) -> List[float]:
    out = []
    for name in LAYERS:
        grads = [p.grad for p in getattr(net, name).parameters() if p.grad is not None]
        out.append(0.0 if name == "fc2" or not grads else float(sum(g.abs().sum() for g in grads) / len(grads)))
    return out


def explained_variance(y_true: torch.Tensor, y_pred: torch.Tensor) -> float:
    """sklearn.metrics.explained_variance_score for one output (critic/learner.py:160-163)."""
    y_true, y_pred = y_true.double().flatten(), y_pred.double().flatten()
    diff = y_true - y_pred
    num = torch.mean((diff - diff.mean()) ** 2)
    den = torch.mean((y_true - y_true.mean()) ** 2)
    if float(den) == 0.0:
        return 1.0 if float(num) == 0.0 else 0.0
    return float(1.0 - num / den)


def critic_metrics(steps: Sequence[Dict[str, torch.Tensor]], critic: torch.nn.Module) -> Dict[str, float]:
    """``steps``: one record per minibatch of the pass (CriticLearner.last): loss, q_chosen [B,1] (pre-step), td [B],
    discounted [B], q_new [B,A] (post-step), logp_chosen [B,1]."""
    loss = torch.stack([s["loss"] for s in steps])
    td = torch.stack([s["td"].flatten() for s in steps])
    dr = torch.stack([s["discounted"].flatten() for s in steps])
    q_chosen = torch.stack([s["q_chosen"].flatten() for s in steps])
    q_all = torch.stack([s["q_new"] for s in steps])
    dev = torch.abs(dr - q_chosen)
    out = {
        "Critic/Loss": float(loss.mean()),
        "Critic/TD-Targets mean": float(td.mean()),
        "Critic/TD-Targets std": float(td.std()),
        "Critic/Q chosen mean": float(q_chosen.mean()),
        "Critic/Q values mean": float(q_all.mean()),
        "Critic/Q values min": float(q_all.min()),
        "Critic/Q values std": float(q_all.std()),
        "Critic/Explained variance": explained_variance(dr, td),
        "Critic/Discounted returns mean": float(dr.mean()),
        "Critic/Discounted_returns std": float(dr.std()),
        "Critic/Abs deviation Q-value <-> Return mean": float(dev.mean()),
        "Critic/Abs deviation Q-value <-> Return std": float(dev.std()),
        "Critic/Log probs according to critic": float(torch.stack([s["logp_chosen"].flatten() for s in steps]).mean()),
    }
    for tag, v in zip(LAYER_TAGS, grad_l1_norms(critic)):
        out[f"Parameters/Critic/{tag} gradients"] = v
    return out


def actor_metrics(steps: Sequence[Dict[str, torch.Tensor]], actor: torch.nn.Module, probs_after: Sequence[torch.Tensor]) -> Dict[str, float]:
    """``steps``: per minibatch (ActorLearner.last): loss, adv [B], log_probs [B,A], hidden0 [256]; ``probs_after``: the
    updated actor's (eps-mixed) probabilities on the same minibatches."""
    log_all = torch.stack([s["log_probs"] for s in steps])           # [nb,B,A]
    p_old = torch.exp(log_all)
    chosen = torch.stack([s["log_chosen"].flatten() for s in steps])
    adv = torch.stack([s["adv"].flatten() for s in steps])
    entropy = -(p_old * torch.log(p_old)).sum(-1)                     # scipy.stats.entropy over the action axis
    q = torch.exp(torch.stack(list(probs_after)))                     # exp of PROBABILITIES, as the script does
    kl = (p_old * torch.log(p_old / q)).sum(0)                        # python sum() over the minibatch axis
    out = {
        "Actor/Loss": float(torch.stack([s["loss"] for s in steps]).mean()),
        "Actor/Advantages mean": float(adv.mean()),
        "Actor/Advantages std": float(adv.std()),
        "Actor/Log probs chosen mean": float(chosen.mean()),
        "Actor/Policy entropy": float(entropy.mean()),
        "Actor/KL divergence policy": float(kl.mean()),
        "Actor/Hidden state entropy": float(torch.stack([torch.mean(s["hidden0"] ** 2) for s in steps]).mean()),
    }
    for tag, v in zip(LAYER_TAGS, grad_l1_norms(actor)):
        out[f"Parameters/Actor/{tag} gradients"] = v
    return out


def return_scalars(mode: str, absolute_returns, episode_rewards, episode_returns) -> Dict[str, float]:
    """The twelve return / reward statistics of add_to_tensorboard (coma_mission.py:208-267); population std like np.std."""
    out = {}
    for name, values in ((f"{mode}Return/Episode", absolute_returns), (f"{mode}Rewards/Episode", episode_rewards),
                         (f"{mode}Return/Relative(used)/Episode", episode_returns)):
        v = torch.as_tensor(values, dtype=torch.float64).flatten()
        out[f"{name}/mean"] = float(v.mean())
        out[f"{name}/std"] = float(v.std(unbiased=False))
        out[f"{name}/max"] = float(v.max())
        out[f"{name}/min"] = float(v.min())
    return out
