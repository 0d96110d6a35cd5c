"""Batched COMA training loop: E device-resident environments roll out in lock step, the update runs on the whole
buffer.  Mirrors the cadence of COMAMission.execute (missions/coma_mission.py:48-172): rollout -> TD(lambda) targets
(K8) -> data_passes x batch_number minibatch steps of critic then actor (K7 inside the actor step).

One "update" = TD-target build + data_passes * batch_number Adam steps of each net, exactly the reference's round; the
number of transitions per round scales with the number of batched envs (reference: 300; here waves * T * E * N).
"""
from __future__ import annotations

import copy
from typing import Dict, Optional

import torch

from . import _ffi
from .learners import ActorLearner, CriticLearner
from .networks import ActorNetwork, CriticNetwork, epsilon_schedule
from .parallel import GradAllReducer, broadcast_module, episode_ids
from .vec_env import VecEnv, POLICY_ARGMAX, POLICY_SAMPLE


class COMATrainer:
    def __init__(self, params: Dict, n_envs: int, device: str = "cuda:0", philox_seed: int = 3, waves_per_update: int = 1,
                 quirks: str = "reference", rank: int = 0, world: int = 1, first_episode: int = 1,
                 terrain: str = "split", graphs: bool = False, placement_draws: int = 0, team_sizes=None):
        self.params = params
        # team_sizes: mixed team sizes in one batch (VecEnv): the transitions of agents that do not fly never enter a minibatch
        self.env = VecEnv(params, n_envs, device=device, philox_seed=philox_seed, terrain=terrain, team_sizes=team_sizes)
        # Large batches: where the allocator put the maps decides ~10 % of the rollout's map kernels (VecEnv.tune_placement).  The
        # search is OPT-IN (placement_draws > 1; bench.py passes its own --placement-draws): it steps the env for ~10 ms a draw and
        # holds candidate arenas within half of the free memory while it runs, which a caller should ask for, not find out about.
        self.placement = self.env.tune_placement(placement_draws) if placement_draws > 1 else None
        self.device = self.env.device
        self.rank, self.world = rank, world
        self.first_episode = first_episode
        self.waves_per_update = waves_per_update
        self.quirks = quirks
        d = self.env.d
        self.T, self.E, self.N, self.A = d.budget + 1, self.env.E, d.n_agents, d.n_actions
        net = params["networks"]
        self.data_passes, self.batch_number = net["data_passes"], net["batch_number"]
        self.actor = ActorNetwork(params)
        self.critic = CriticNetwork(params)
        # the reference builds TD targets with a deepcopy of the critic taken at construction that is
        # never synchronised; quirks="fixed" uses the learner's synchronised target network instead
        self.frozen_target = copy.deepcopy(self.critic).to(self.device).eval()
        # graphs: the round is launch-bound at small env counts (the reference's own 5 episodes per round: ~3000 launches of a
        # few microseconds of work each, Python between them); capture_graphs() records every rollout step and the whole update
        # into hipGraphs.  graphs=True changes the optimizer IMPLEMENTATION for every round of this trainer, recorded or not: both
        # Adam instances are torch's fused capturable form (step counters on the device, one kernel per step), whose update agrees
        # with the default form to float32 rounding, not bit for bit; their state is not part of save_actor's pickle.
        self.graphs = bool(graphs)
        self._step_graphs = self._update_graph = None
        self.actor_learner = ActorLearner(params, self.actor, self.device, ctx=self.env.ctx, capturable=self.graphs)
        self.critic_learner = CriticLearner(params, self.critic, self.device, capturable=self.graphs)
        for m in (self.actor, self.critic, self.critic_learner.target_critic, self.frozen_target):
            broadcast_module(m)
        # gradients of both nets live in one flat buffer (critic, then actor): the data-parallel average is one all-reduce of a
        # view, and the actor's gradients of minibatch b travel together with the critic's of minibatch b+1
        self.reducer = GradAllReducer().attach(self.critic, self.actor, skip=GradAllReducer.unused_fc2)
        self._grads_checked = False
        W, T, E, N = waves_per_update, self.T, self.E, self.N
        dev = self.device
        self.buf_obs = torch.empty(W, T, E, N, 11, 11, 7, device=dev)
        self.buf_state = torch.empty(W, T, E, N, 11, 11, 12, device=dev)
        self.buf_action = torch.empty(W, T, E, N, dtype=torch.int32, device=dev)
        self.buf_mask = torch.empty(W, T, E, N, self.A, dtype=torch.uint8, device=dev)
        self.buf_reward = torch.empty(W, T, E, device=dev)
        self.wave = 0          # rollout waves done so far (defines the episode numbers)
        self.filled = 0        # waves currently in the buffer
        self.train_step = 0
        self.eps = epsilon_schedule(params, 0)
        self.eps_dev = torch.full((), float(self.eps), dtype=torch.float32, device=dev)   # epsilon as the graphs read it
        self.ret = torch.zeros(E, device=dev)
        self.abs_ret = torch.zeros(E, device=dev)
        self.keep_rollout_log = False
        self.last_rollout = None
        self.last_diagnostics = None

    # ------------------------------------------------------------------------------------------------
    def rollout(self, mode: str = "train") -> Dict[str, float]:
        """One wave: E episodes in lock step (EpisodeGenerator.execute for every env at once)."""
        env = self.env
        eps_ids = episode_ids(self.first_episode, self.wave, self.E, self.rank, self.world)
        # the reference anneals epsilon with the episode index (actor/network.py:53-58); one wave = E episodes
        self.eps = epsilon_schedule(self.params, int(eps_ids[0]))
        self.eps_dev.fill_(float(self.eps))
        env.reset(eps_ids)
        w = self.filled
        ret, abs_ret = self.ret.zero_(), self.abs_ret.zero_()
        policy = POLICY_ARGMAX if mode == "eval" else POLICY_SAMPLE
        step_rewards, step_actions, step_altitudes = [], [], []
        replay = self._step_graphs is not None and mode == "train" and w == 0 and not self.keep_rollout_log
        for t in range(self.T):
            if replay:
                self._step_graphs[t].replay()
                # (the host-side bookkeeping of the env that a replayed step skips: the step counter, no fusion pending, the
                #  observations held are those of step t)
                env.t, env._pending_t, env._obs_t = t + 1, None, t
                continue
            reward = self._rollout_step(t, w, policy, mode == "train", self.eps_dev if self.graphs else self.eps)
            if self.keep_rollout_log:
                step_rewards.append(reward[:, 0].clone())
                step_actions.append(env.action.clone())
                step_altitudes.append(env.pos[:, :, 2].clone())
        if self.keep_rollout_log:   # per-episode figures for the mission log (coma_mission.py:78-82)
            self.last_rollout = dict(episode_returns=ret.clone(), absolute_returns=abs_ret.clone(),
                                     rewards=torch.stack(step_rewards, 1), actions=torch.stack(step_actions, 1),
                                     altitudes=torch.stack(step_altitudes, 1))
        self.wave += 1
        if mode == "train":
            self.filled += 1
        return {"episode_return": float(ret.mean()), "absolute_return": float(abs_ret.mean()), "eps": self.eps,
                "faults": int(env.fault.ne(0).sum())}

    def _rollout_step(self, t: int, w: int, policy: int, store: bool, eps):
        """One lock-step env step of all E envs: observations -> actor -> move + sense (+ the transition into the buffer)."""
        env = self.env
        obs = env.build_observations(t)
        with torch.no_grad():
            probs, _ = self.actor(obs.view(self.E * self.N, 11, 11, 7), eps)
        reward, done, state = env.steps(t, policy=policy, probs=probs.view(self.E, self.N, self.A))
        if store:
            self.buf_obs[w, t].copy_(obs)
            self.buf_state[w, t].copy_(state)
            self.buf_action[w, t].copy_(env.action)
            self.buf_mask[w, t].copy_(env.mask)
            self.buf_reward[w, t].copy_(reward[:, 0])
        self.ret += reward[:, 0]
        self.abs_ret += reward[:, 1]
        return reward

    def capture_graphs(self):
        """Records the launch-bound round into hipGraphs: one graph per rollout step t of a training wave (the step number is an
        argument of several kernels, everything else -- episode numbers, epsilon, every array -- is read from fixed device
        buffers) and one graph for the whole update (TD targets + data_passes x batch_number minibatch steps of both nets, the
        minibatch permutations read from a fixed buffer that update() refills).  Needs one eager round first (MIOpen's kernel
        selection, allocator warm-up); waves_per_update = 1, hard target updates, one rank."""
        if not self.graphs:
            raise _ffi.IppmError("COMATrainer(graphs=True) is needed: captured optimizer steps keep their counters on the device")
        if self.waves_per_update != 1 or self.world != 1 or self.critic_learner.target_update_mode != "hard" or self.env.n_active is not None:
            raise _ffi.IppmError("capture_graphs: one wave per update, one rank, hard target updates and one team size only")
        env = self.env
        env.profile = False
        torch.cuda.synchronize(self.device)
        saved = {k: getattr(env, k).clone() for k in ("local", "glob", "ws", "sums", "pos", "pos_pre", "comm", "mask", "action", "fault",
                                                       "reward", "area", "rect", "rect_next", "code", "work")}
        stream = torch.cuda.Stream(device=self.device)
        graphs = []
        for t in range(self.T):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                self._rollout_step(t, 0, POLICY_SAMPLE, True, self.eps_dev)
            graphs.append(g)
        n = self.T * self.E * self.N
        self._perms = torch.stack([torch.randperm(n, device=self.device) for _ in range(self.data_passes)])
        self._loss_out = torch.zeros(2, device=self.device)
        filled, self.filled = self.filled, 1
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            closs, aloss = self._update_compute(self._perms, self.eps_dev, False)
            self._loss_out[0].copy_(closs)
            self._loss_out[1].copy_(aloss)
        self.filled = filled
        torch.cuda.synchronize(self.device)
        for k, v in saved.items():   # capture does not execute; keep the env's state untouched in any case
            getattr(env, k).copy_(v)
        self._step_graphs, self._update_graph = graphs, g
        env._graphs = graphs            # (the env's buffers are baked into the recording: VecEnv.tune_placement refuses from here on)

    # ------------------------------------------------------------------------------------------------
    def td_targets(self):
        """K8 over one chain per (env, agent): the transitions of all buffered waves in time order
        (BatchMemory.build_td_targets, batch_memory.py:120-162)."""
        W, T, E, N = self.filled, self.T, self.E, self.N
        target = self.frozen_target if self.quirks == "reference" else self.critic_learner.target_critic
        states = self.buf_state[:W].reshape(W * T * E * N, 11, 11, 12)
        actions = self.buf_action[:W].reshape(-1)
        q_sel = torch.empty(W * T * E * N, device=self.device)
        with torch.no_grad():
            chunk = 16384
            for lo in range(0, states.shape[0], chunk):
                q, _ = target(states[lo:lo + chunk])
                q_sel[lo:lo + chunk] = q.view(-1, self.A).gather(1, actions[lo:lo + chunk].long().view(-1, 1)).squeeze(1)
        # chains: [E*N, W*T]
        q_sel = q_sel.view(W, T, E, N).permute(2, 3, 0, 1).reshape(E * N, W * T).contiguous()
        rew = self.buf_reward[:W].unsqueeze(-1).expand(W, T, E, N).permute(2, 3, 0, 1).reshape(E * N, W * T).contiguous()
        done = torch.zeros(W, T, dtype=torch.uint8, device=self.device)
        done[:, T - 1] = 1
        done = done.view(1, W * T).expand(E * N, W * T).contiguous()
        td = torch.empty_like(rew)
        dr = torch.empty_like(rew)
        self.env.ctx.call("ippm_td_lambda", _ffi.ptr(rew), _ffi.ptr(done), _ffi.ptr(q_sel), _ffi.ptr(td), _ffi.ptr(dr), E * N,
                          W * T, self.env.stream)
        # back to buffer order [W,T,E,N]
        to_buf = lambda x: x.view(E, N, W, T).permute(2, 3, 0, 1).reshape(-1)  # noqa: E731
        return to_buf(td), to_buf(dr)

    def update(self, diagnostics: bool = False) -> Dict[str, float]:
        """One COMA round on everything in the buffer, then clear it (coma_mission.py:89-116).  ``diagnostics``: also
        compute the reference's per-training-step TensorBoard figures from data pass 0 (ippmarl.metrics) into
        ``self.last_diagnostics`` (costs one extra actor forward per minibatch)."""
        W, T, E, N = self.filled, self.T, self.E, self.N
        assert W > 0, "update() needs at least one rollout wave"
        n = W * T * E * N
        if self._update_graph is not None and W == 1 and not diagnostics:
            # the recorded round: hard target copy (a host-side decision) first, fresh permutations into the graph's buffer, replay
            self.critic_learner.update_target_network(self.train_step, 0)
            for dp in range(self.data_passes):
                self._perms[dp].copy_(torch.randperm(n, device=self.device))
            self._update_graph.replay()
            self.train_step += 1
            self.filled = 0
            closs, aloss = self._loss_out.tolist()
            return {"critic_loss": closs, "actor_loss": aloss, "transitions": n * self.world,
                    "adam_steps": 2 * self.data_passes * self.batch_number, "train_step": self.train_step}
        closs, aloss = self._update_compute(None, self.eps_dev if self.graphs else self.eps, diagnostics)
        self.filled = 0
        total = n * self.world
        if self.env.n_active is not None:
            # mixed team sizes: every rank counts its own flying agents (the ranks' teams differ), summed over the ranks
            mine = torch.tensor([int(self.env.n_active.sum()) * W * T], dtype=torch.int64, device=self.device)
            if self.world > 1:
                import torch.distributed as dist
                if dist.get_backend() == "gloo":
                    mine = mine.cpu()
                dist.all_reduce(mine)
            total = int(mine[0])
        return {"critic_loss": float(closs), "actor_loss": float(aloss), "transitions": total,
                "adam_steps": 2 * self.data_passes * self.batch_number, "train_step": self.train_step}

    def valid_transitions(self, waves: int) -> Optional[torch.Tensor]:
        """Indices into the flattened [waves, T, E, N] buffers of the transitions of agents that fly (None: all of them do)."""
        if self.env.n_active is None:
            return None
        flying = torch.arange(self.N, device=self.device).view(1, self.N) < self.env.n_active.view(self.E, 1)     # [E, N]
        return flying.view(1, 1, self.E, self.N).expand(waves, self.T, self.E, self.N).reshape(-1).nonzero().view(-1)

    def _update_compute(self, perms, eps, diagnostics: bool):
        """The arithmetic of one round.  ``perms`` None: eager (draws the minibatch permutations, syncs the target network and
        counts train steps as it goes); a [data_passes, n] tensor: the form capture_graphs records (those host-side pieces are
        done by update() around the replay)."""
        W, T, E, N = self.filled, self.T, self.E, self.N
        n = W * T * E * N
        eager = perms is None
        td, dr = self.td_targets()
        obs = self.buf_obs[:W].reshape(n, 11, 11, 7)
        states = self.buf_state[:W].reshape(n, 11, 11, 12)
        actions = self.buf_action[:W].reshape(n)
        masks = self.buf_mask[:W].reshape(n, self.A)
        flying = self.valid_transitions(W)           # mixed team sizes: minibatches are drawn from the agents that fly
        nv = n if flying is None else int(flying.numel())
        bs = nv // self.batch_number
        closs = aloss = torch.zeros((), device=self.device)
        for data_pass in range(self.data_passes):
            perm = torch.randperm(nv, device=self.device) if eager else perms[data_pass]
            if flying is not None:
                perm = flying[perm]
            if eager:
                self.critic_learner.update_target_network(self.train_step, data_pass)
            collect = diagnostics and data_pass == 0
            self.critic_learner.collect = self.actor_learner.collect = collect
            crit_rec, act_rec = [], []
            # Reference order (critic/learner.py:58-105, actor/learner.py:36-101): all critic minibatches, each followed by
            # the post-step Q of ITS minibatch, then all actor minibatches with those Q.  Neither net's step feeds the
            # other's, so interleaving them -- actor(b) right after critic(b) -- computes the same numbers; it lets the
            # gradient exchange of actor(b) and critic(b+1) share one all-reduce (B+1 collectives per pass, not 2B).
            idx = [perm[b * bs:(b + 1) * bs] for b in range(self.batch_number)]
            closs = self.critic_learner.backward(states[idx[0]], actions[idx[0]], td[idx[0]])
            if not self._grads_checked and eager:   # once: no gradient of the attached nets lives outside the reduced buffer
                self.reducer.check_covered()
            self.reducer(self.critic)
            q_b = self.critic_learner.apply()
            for b in range(self.batch_number):
                if collect:
                    crit_rec.append(dict(self.critic_learner.last, discounted=dr[idx[b]]))
                aloss, _ = self.actor_learner.backward(obs[idx[b]], actions[idx[b]], masks[idx[b]], q_b, eps)
                if collect:
                    act_rec.append(self.actor_learner.last)
                if not self._grads_checked and eager:
                    self.reducer.check_covered()
                    self._grads_checked = True
                if b + 1 < self.batch_number:
                    closs = self.critic_learner.backward(states[idx[b + 1]], actions[idx[b + 1]], td[idx[b + 1]])
                    self.reducer(self.critic, self.actor)
                    self.actor_learner.apply()
                    q_b = self.critic_learner.apply()
                else:
                    self.reducer(self.actor)
                    self.actor_learner.apply()
            if collect:
                from . import metrics
                with torch.no_grad():
                    after = [self.actor(obs[perm[b * bs:(b + 1) * bs]], eps)[0] for b in range(self.batch_number)]
                self.last_diagnostics = dict(metrics.critic_metrics(crit_rec, self.critic))
                self.last_diagnostics.update(metrics.actor_metrics(act_rec, self.actor, after))
                self.critic_learner.collect = self.actor_learner.collect = False
            if data_pass == 0 and eager:
                self.train_step += 1
        return closs, aloss

    def train(self, n_updates: int, log=None):
        history = []
        for _ in range(n_updates):
            stats = {}
            for _ in range(self.waves_per_update):
                stats = self.rollout("train")
            stats.update(self.update())
            history.append(stats)
            if log:
                log(stats)
        return history

    # ------------------------------------------------------------------------------------------------
    def map_metrics(self, glob: Optional[torch.Tensor] = None):
        """Per-env evaluation metrics of fused global maps (default: the env's; coma_test.py:84-97,177-196;
        IG_baseline.py:84-97): mean entropy over the target cells (weights from the ground truth) and F1 of the target
        class at p > 0.5."""
        env = self.env
        glob = env.glob if glob is None else glob
        ent = torch.zeros(self.E, dtype=torch.float64, device=self.device)
        env.ctx.call("ippm_weighted_entropy", env._p(glob), env._p(env.truth), 1, _ffi.ptr(ent), self.E, env.stream)
        counts = self.f1_counts(glob, 0.0)
        tp, fp, fn = counts[:, 0].double(), counts[:, 1].double(), counts[:, 2].double()
        target = (tp + fn).clamp_min(1)
        f1 = torch.where(2 * tp + fp + fn > 0, 2 * tp / (2 * tp + fp + fn).clamp_min(1), torch.zeros_like(tp))
        return ent / target, f1

    def f1_counts(self, glob: Optional[torch.Tensor] = None, logodds_threshold: float = 0.0) -> torch.Tensor:
        """int64 [E,3] = (tp, fp, fn) of the target class for maps thresholded at log-odds > ``logodds_threshold`` (0 <=> the
        reference's p > 0.5, utils/utils.py:64-76).  Thresholds +-1e-5 leave out / take in the exactly-cancelled cells, whose class
        is rounding noise in the reference: the counts under those two are exact integers."""
        env = self.env
        glob = env.glob if glob is None else glob
        counts = torch.zeros(self.E, 3, dtype=torch.int64, device=self.device)
        env.ctx.call("ippm_f1_counts", env._p(glob), env._p(env.truth), 1, float(logodds_threshold), _ffi.ptr(counts), self.E, env.stream)
        return counts

    def global_map_with_pending(self) -> torch.Tensor:
        """The global maps with the measurements of the CURRENT positions fused in, as log-odds [E,gx,gy], without touching
        the env.  The env's own global fusion (K5) lags the sensing by one step (it fuses what was published
        before the move); the deployment scripts score the map after fusing the fresh measurements
        (coma_test.py:150-196: ``fuse_map(current_global_map, maps2communicate_list)`` of the new positions)."""
        env = self.env
        glob, ws, sums = env.glob.clone(), env.ws.clone(), env.sums.clone()
        reward = torch.empty_like(env.reward)
        env.ctx.call("ippm_fuse_global_reward", _ffi.ptr(glob), env._p(env.code), env._p(env.rect), env._p(env.pos), _ffi.ptr(ws),
                     _ffi.ptr(sums), _ffi.ptr(reward), self.E, env.stream)
        return glob

    def evaluate(self, waves: int = 1, counts_log: Optional[list] = None) -> Dict[str, object]:
        """Greedy (argmax) deployment of the current actor, the reference's coma_test loop for E envs at once: mean return and
        the per-step curves of target entropy and F1.  Index 0 = the prior map, index t+1 = the map holding every
        measurement up to and including the sensing of step t (coma_test.py:84-97,150-196).  ``counts_log`` (a list) receives, per
        scored map, the pair of int64 [E,3] count tensors of f1_counts at log-odds thresholds +1e-5 and -1e-5."""
        returns, ent_curves, f1_curves = [], [], []

        def log_counts(glob=None):
            if counts_log is not None:
                counts_log.append((self.f1_counts(glob, 1e-5).cpu(), self.f1_counts(glob, -1e-5).cpu()))

        for _ in range(waves):
            env = self.env
            eps_ids = episode_ids(self.first_episode, self.wave, self.E, self.rank, self.world)
            env.reset(eps_ids)
            e0, f0 = self.map_metrics()   # reset senses at the start cells, the global map is still the prior
            log_counts()
            ents, f1s = [e0.mean().item()], [f0.mean().item()]
            ret = torch.zeros(self.E, device=self.device)
            for t in range(self.T):
                obs = env.build_observations(t)
                with torch.no_grad():
                    probs, _ = self.actor(obs.view(self.E * self.N, 11, 11, 7), self.eps)
                reward, _, _ = env.steps(t, policy=POLICY_ARGMAX, probs=probs.view(self.E, self.N, self.A))
                ret += reward[:, 0]
                pending = self.global_map_with_pending()
                e, f = self.map_metrics(pending)
                log_counts(pending)
                ents.append(e.mean().item())
                f1s.append(f.mean().item())
            self.wave += 1
            returns.append(float(ret.mean()))
            ent_curves.append(ents)
            f1_curves.append(f1s)
        mean = lambda rows: [sum(c) / len(c) for c in zip(*rows)]  # noqa: E731
        return {"episode_return": sum(returns) / len(returns), "target_entropy": mean(ent_curves), "f1": mean(f1_curves)}

    def returns_on(self, episodes, policy: str = "actor") -> Dict[str, float]:
        """Mean return of ``policy`` over the FIXED ``episodes`` (E ids: same truth, start cells and sensor noise whoever flies them --
        every random stream is keyed by the episode number), the yardstick of the reference's own comparison (coma_test.py:84-97,
        random_baseline.py:91-96, IG_baseline.py:127-148):
          "actor"   the current actor, greedy (argmax of probs * mask: ActorNetwork in "eval" mode, actor/network.py:63-66)
          "random"  uniform over the valid actions
          "ig"      the greedy expected-information-gain planner on the agents' local maps (one team size per context)
        Does not touch the training buffer, the wave counter or epsilon.
        -> mean relative return ("episode_return": what COMA's reward is made of, utils/reward.py:25-40), mean absolute return, and
        the final global maps' mean target-region entropy and F1."""
        from .vec_env import POLICY_EXPLICIT, POLICY_UNIFORM
        if policy not in ("actor", "random", "ig"):
            raise ValueError(f"unknown policy {policy!r}")
        env = self.env
        env.reset(torch.as_tensor(episodes, dtype=torch.int64).reshape(self.E))
        ret = torch.zeros(self.E, device=self.device)
        abs_ret = torch.zeros(self.E, device=self.device)
        for t in range(self.T):
            if policy == "actor":
                obs = env.build_observations(t)
                with torch.no_grad():
                    probs, _ = self.actor(obs.view(self.E * self.N, 11, 11, 7), self.eps_dev if self.graphs else self.eps)
                reward, _, _ = env.steps(t, policy=POLICY_ARGMAX, probs=probs.view(self.E, self.N, self.A), features=False)
            elif policy == "ig":
                env.build_observations(t, features=False)
                reward, _, _ = env.steps(t, policy=POLICY_EXPLICIT, actions=env.ig_actions(communication=True), features=False)
            else:
                env.build_observations(t, features=False)
                reward, _, _ = env.steps(t, policy=POLICY_UNIFORM, features=False)
            ret += reward[:, 0]
            abs_ret += reward[:, 1]
        ent, f1 = self.map_metrics(self.global_map_with_pending())
        return {"episode_return": float(ret.mean()), "absolute_return": float(abs_ret.mean()), "return_std": float(ret.std()),
                "final_target_entropy": float(ent.mean()), "final_f1": float(f1.mean()), "faults": int(env.fault.ne(0).sum())}

    def save_actor(self, path: str):
        """Whole-module pickle of the actor, the reference's checkpoint format (coma_mission.py:425-451)."""
        from .checkpoint import save_actor
        save_actor(self.actor, path)
