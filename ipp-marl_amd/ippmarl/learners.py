"""COMA learners on tensors: critic TD regression (critic/learner.py:58-198) and actor policy gradient with the
counterfactual baseline (actor/learner.py:36-168).  The baseline/advantage arithmetic runs in the HIP kernel K7
(ippm_coma_advantage); autograd, Adam and the convnets are PyTorch.  Metric-only work of the reference (KL via an
extra forward pass, gradient-norm dumps, explained variance) is computed by ippmarl.metrics from the per-minibatch
records the learners keep in ``.last`` when ``collect`` is switched on."""
from __future__ import annotations

import copy
from typing import Dict, Optional

import torch

from . import _ffi


def _adam(parameters, lr: float, capturable: bool):
    """Adam as the reference configures it (actor/learner.py:27, critic/learner.py:37).  ``capturable`` (hipGraph rounds): the
    step counter lives on the device and the whole update of a net is ONE fused kernel instead of ~10 multi-tensor launches per
    step -- the recorded round is nothing but small launches, so their number is its duration."""
    if capturable:
        return torch.optim.Adam(parameters, lr=lr, capturable=True, fused=True)
    return torch.optim.Adam(parameters, lr=lr)


class CriticLearner:
    def __init__(self, params: Dict, critic, device, capturable: bool = False):
        self.params = params
        self.critic = critic.to(device)
        self.device = device
        self.target_critic = copy.deepcopy(critic).to(device)
        self.target_critic.eval()
        net = params["networks"]
        self.copy_rate = net["copy_rate"]
        self.lr = net["critic"]["learning_rate"]
        self.target_update_mode = net["critic"]["target_update_mode"]
        self.tau = net["critic"]["tau"]
        # capturable: the step count lives on the device, so that optimizer steps can be recorded into a hipGraph (trainer.capture_graphs)
        self.optimizer = _adam(self.critic.parameters(), self.lr, capturable)
        self.optimizer.zero_grad()
        self.collect = False   # keep the tensors ippmarl.metrics.critic_metrics needs in self.last
        self.last = None

    def update_target_network(self, num_train_step: int, data_pass: int):
        """Hard copy every copy_rate train steps on data pass 0, or Polyak (critic/learner.py:192-198)."""
        if self.target_update_mode == "hard":
            if num_train_step % self.copy_rate == 0 and data_pass == 0:
                self.target_critic.load_state_dict(self.critic.state_dict())
        elif self.target_update_mode == "soft":
            with torch.no_grad():
                for t, s in zip(self.target_critic.parameters(), self.critic.parameters()):
                    t.mul_(1 - self.tau).add_(s, alpha=self.tau)

    def backward(self, states: torch.Tensor, actions: torch.Tensor, td_targets: torch.Tensor):
        """First half of a minibatch step: MSE on the chosen Q and its gradients (critic/learner.py:76-92).  The gradients
        are cleared in place, so views handed out by parallel.GradAllReducer.attach stay valid."""
        q, _ = self.critic(states)
        q_chosen = q.gather(1, actions.long().view(-1, 1))
        loss = torch.square(q_chosen - td_targets.view(-1, 1).detach()).squeeze().mean()
        self.optimizer.zero_grad(set_to_none=False)
        loss.backward()
        self._pending = (states, actions, td_targets, loss.detach(), q_chosen.detach())
        return loss.detach()

    def apply(self):
        """Second half: Adam step, then the POST-step Q handed to the actor (critic/learner.py:93-105).  -> q_new [B,A]"""
        states, actions, td_targets, loss, q_chosen = self._pending
        self._pending = None
        self.optimizer.step()
        with torch.no_grad():
            q_new, logp = self.critic(states)
        if self.collect:
            self.last = dict(loss=loss, q_chosen=q_chosen, td=td_targets.detach(), q_new=q_new,
                             logp_chosen=logp.view(-1, q_new.shape[-1]).gather(1, actions.long().view(-1, 1)))
        return q_new

    def step(self, states: torch.Tensor, actions: torch.Tensor, td_targets: torch.Tensor, grad_hook=None):
        """One minibatch: MSE on the chosen Q, Adam step, then the POST-step Q handed to the actor
        (critic/learner.py:76-105).  Returns (loss, q_new [B,A])."""
        loss = self.backward(states, actions, td_targets)
        if grad_hook is not None:
            grad_hook(self.critic)
        return loss, self.apply()


class ActorLearner:
    def __init__(self, params: Dict, actor, device, ctx: Optional[_ffi.Context] = None, capturable: bool = False):
        self.params = params
        self.actor = actor.to(device)
        self.device = device
        self.ctx = ctx
        self.n_actions = params["experiment"]["constraints"]["num_actions"]
        self.lr = params["networks"]["actor"]["learning_rate"]
        self.optimizer = _adam(self.actor.parameters(), self.lr, capturable)
        self.optimizer.zero_grad()
        self.collect = False
        self.last = None

    def advantage(self, probs: torch.Tensor, q_values: torch.Tensor, masks: torch.Tensor, actions: torch.Tensor):
        """A = Q(a) - sum_a' pi~(a') Q(a') mask(a') with pi~ the mask-renormalised, floored policy
        (actor/learner.py:55-83) -- HIP kernel K7."""
        if self.ctx is None:
            raise _ffi.IppmError("ActorLearner needs an ippmarl context: the counterfactual baseline runs in libippmarl (K7)")
        b = probs.shape[0]
        adv = torch.empty(b, dtype=torch.float32, device=probs.device)
        # converted copies must stay referenced until the launch is queued (a dropped temporary's block can be reused)
        p32 = probs.detach().float().contiguous()
        q32 = q_values.float().contiguous()
        m8 = masks.to(torch.uint8).contiguous()
        a32 = actions.to(torch.int32).contiguous()
        stream = torch.cuda.current_stream(probs.device).cuda_stream
        self.ctx.call("ippm_coma_advantage", _ffi.ptr(p32), _ffi.ptr(q32), _ffi.ptr(m8), _ffi.ptr(a32), _ffi.ptr(adv), None, b, stream)
        return adv

    def backward(self, observations: torch.Tensor, actions: torch.Tensor, masks: torch.Tensor, q_values: torch.Tensor, eps: float):
        """First half of a minibatch step (actor/learner.py:52-97): loss and gradients.  The reference's loss broadcasts
        [B,1]*[B,1]*[B,A] before the mean, i.e. every sample is weighted by (#valid actions)/A."""
        probs, hidden = self.actor(observations, eps)
        log_probs = torch.log(probs)
        adv = self.advantage(probs, q_values, masks, actions)
        log_chosen = log_probs.gather(1, actions.long().view(-1, 1)).squeeze(1)
        weight = masks.to(log_chosen.dtype).sum(-1) / self.n_actions
        loss = -(adv.detach() * log_chosen * weight).mean()
        self.optimizer.zero_grad(set_to_none=False)
        loss.backward()
        if self.collect:
            self.last = dict(loss=loss.detach(), adv=adv.detach(), log_probs=log_probs.detach(), log_chosen=log_chosen.detach(),
                             hidden0=hidden[0].detach())
        return loss.detach(), adv

    def apply(self):
        self.optimizer.step()

    def step(self, observations: torch.Tensor, actions: torch.Tensor, masks: torch.Tensor, q_values: torch.Tensor, eps: float,
             grad_hook=None):
        """One minibatch (actor/learner.py:52-101)."""
        loss, adv = self.backward(observations, actions, masks, q_values, eps)
        if grad_hook is not None:
            grad_hook(self.actor)
        self.apply()
        return loss, adv
