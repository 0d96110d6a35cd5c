"""COMA training mission with the reference's class name, cadence and TensorBoard scalar names
(missions/coma_mission.py:24-451), driven by the batched trainer.

The reference plays one episode at a time and updates whenever its memory holds batch_size * batch_number transitions
(5 episodes at 4 UAVs and budget 14); it runs ``n_episodes`` such updates, evaluates 50 greedy episodes every 50
updates, logs ~45 scalars per update and keeps the actor with the best running mean return.  Here one rollout *wave* of
``n_envs`` lock-step episodes feeds each update; ``n_envs`` defaults to the reference's episodes-per-update so that an
update sees the same number of transitions, and can be raised to thousands (that is the point of the GPU path).
Bar-plot figures of the reference (sampled actions / altitudes) are kept as counts in ``last_counts`` instead."""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch

from .. import metrics
from ..checkpoint import save_actor
from ..trainer import COMATrainer
from .missions import Mission


class COMAMission(Mission):
    def __init__(self, params: Dict, writer, max_mean_episode_return: float = -100, n_envs: Optional[int] = None,
                 log_dir: str = "logs", device: str = "cuda:0", eval_every: int = 50, eval_episodes: int = 50, **trainer_kwargs):
        super().__init__(params, writer, max_mean_episode_return)
        self.num_episodes = params["experiment"]["missions"]["n_episodes"]
        self.batch_size = params["networks"]["batch_size"]
        self.batch_number = params["networks"]["batch_number"]
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.n_actions = params["experiment"]["constraints"]["num_actions"]
        self.budget = params["experiment"]["constraints"]["budget"]
        self.data_passes = params["networks"]["data_passes"]
        self.patience = params["experiment"]["missions"]["patience"]
        per_episode = (self.budget + 1) * self.n_agents
        self.n_envs = n_envs or max(1, math.ceil(self.batch_size * self.batch_number / per_episode))
        self.trainer = COMATrainer(params, self.n_envs, device=device, **trainer_kwargs)
        self.trainer.keep_rollout_log = True
        self.log_dir = log_dir
        self.eval_every, self.eval_episodes = eval_every, eval_episodes
        self.training_step_idx = 0
        self.environment_step_idx = 0
        self.episode_returns = []
        self.mode = "train"
        self.last_counts = None

    # ------------------------------------------------------------------------------------------------
    def add_to_tensorboard(self, rollouts, diagnostics=None):
        """coma_mission.py:174-424: return / reward statistics of the episodes since the last log (+ learner diagnostics in
        train mode), under the reference's tags."""
        cat = lambda key: torch.cat([r[key].flatten() for r in rollouts]).cpu()  # noqa: E731
        actions, altitudes = cat("actions"), cat("altitudes")
        self.last_counts = {"actions": [int((actions == a).sum()) for a in range(self.n_actions)],
                            "altitudes": {int(z): int((altitudes == z).sum()) for z in torch.unique(altitudes)}}
        scalars = metrics.return_scalars(self.mode, cat("absolute_returns"), cat("rewards"), cat("episode_returns"))
        if self.mode == "train" and diagnostics:
            scalars.update(diagnostics)
        if self.writer is not None:
            for tag, value in scalars.items():
                self.writer.add_scalar(tag, value, self.training_step_idx)
        return scalars

    def save_best_model(self, actor_network):
        """coma_mission.py:425-451: keep the actor whenever the running mean return improves (after ``patience`` updates),
        plus snapshots at 300 / 400 / 500 / 600 updates; whole-module pickles like the reference's."""
        running = sum(self.episode_returns) / len(self.episode_returns)
        os.makedirs(self.log_dir, exist_ok=True)
        if len(self.episode_returns) >= self.patience and running > self.max_mean_episode_return:
            self.max_mean_episode_return = running
            save_actor(actor_network, os.path.join(self.log_dir, "best_model.pth"))
        if self.training_step_idx in (300, 400, 500, 600):
            save_actor(actor_network, os.path.join(self.log_dir, f"best_model_{self.training_step_idx}.pth"))

    def execute(self):
        tr = self.trainer
        for _ in range(self.num_episodes):
            rollouts = []
            for _ in range(tr.waves_per_update):
                tr.rollout("train")
                rollouts.append(tr.last_rollout)
            stats = tr.update(diagnostics=self.writer is not None)
            self.training_step_idx = tr.train_step
            self.environment_step_idx += stats["transitions"]
            self.add_to_tensorboard(rollouts, tr.last_diagnostics)
            self.episode_returns.append(float(rollouts[-1]["episode_returns"].mean()))
            self.save_best_model(tr.actor)
            if self.eval_every and self.training_step_idx % self.eval_every == 0:
                self.mode = "eval"
                evals = []
                for _ in range(max(1, math.ceil(self.eval_episodes / self.n_envs))):
                    tr.rollout("eval")
                    evals.append(tr.last_rollout)
                self.add_to_tensorboard(evals)
                self.mode = "train"
        return self.max_mean_episode_return
