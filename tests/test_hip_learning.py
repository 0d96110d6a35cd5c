"""GPU checks of the learning path: the actor's minibatch step (K7 inside) against the reference's recorded numbers,
TD(lambda) chains built from the trainer's buffers, and an end-to-end batched COMA round."""
import numpy as np
import pytest

import ipp_oracle as O
from configs import make_params, synthetic_minibatch

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_actor_and_critic_step_match_reference(golden):
    from ippmarl import _ffi
    from ippmarl.derived import DerivedConstants
    from ippmarl.learners import ActorLearner, CriticLearner
    from ippmarl.networks import ActorNetwork, CriticNetwork
    fx = golden("coma_step")
    params = make_params("c2")
    dev = torch.device("cuda:0")
    torch.manual_seed(int(fx["net_seed"]))
    actor, critic = ActorNetwork(params), CriticNetwork(params)
    ctx = _ffi.Context(DerivedConstants(params))
    obs, state, actions, masks, td = synthetic_minibatch(60, 6, int(fx["mb_seed"]))
    t = lambda x, dt=None: torch.tensor(x, device=dev, dtype=dt)  # noqa: E731
    cl = CriticLearner(params, critic, dev)
    al = ActorLearner(params, actor, dev, ctx=ctx)
    cl.collect = al.collect = True
    closs, q_new = cl.step(t(state), t(actions), t(td))
    np.testing.assert_allclose(float(closs), float(fx["critic_loss"]), rtol=1e-4)
    np.testing.assert_allclose(q_new.cpu().numpy(), fx["q_new"], rtol=2e-4, atol=2e-6)
    # ... and against the same step evaluated in float64 (test_learning_cpu.float64_critic_step: within 1e-6 of the reference's
    # recording, so this deviation is the device path's own -- MIOpen's float32 summation order in three convolutions and their
    # gradients, then Adam's g / (sqrt(v) + eps), which turns a last-bit difference of a tiny gradient into a step of +-lr)
    from test_learning_cpu import float64_critic_step, scale_deviation
    loss64, q64 = float64_critic_step(fx)
    assert abs(float(closs) - loss64) <= 1e-5 * abs(loss64), (float(closs), loss64)
    dev_q = scale_deviation(q_new.cpu().numpy(), q64)
    assert dev_q < 1e-4, dev_q
    aloss, adv = al.step(t(obs, torch.float32), t(actions), t(masks, torch.float32), q_new, float(fx["eps"]))
    np.testing.assert_allclose(float(aloss), float(fx["actor_loss"]), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(float(adv.mean()), float(fx["adv_mean"]), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(float(adv.std()), float(fx["adv_std"]), rtol=2e-4)
    # K7 on the reference's own tensors of this step (its pre-update policy pi0 and post-critic-step Q): here no network
    # runs on the device, so the kernel itself is held to the 1e-5 of the contract -- per sample against the oracle's
    # restatement of actor/learner.py:55-83, and in the two moments the reference recorded
    adv_k7 = al.advantage(t(fx["pi0"]), t(fx["q_new"]), t(masks, torch.float32), t(actions))
    adv_ref, _, _ = O.coma_advantage(fx["pi0"], fx["q_new"], masks.astype(np.float32), actions)
    np.testing.assert_allclose(adv_k7.cpu().numpy(), adv_ref, rtol=1e-5, atol=1e-7)
    # (the mean is a sum of cancelling terms, 26x smaller than the spread: 1e-5 of the advantages' scale)
    np.testing.assert_allclose(float(adv_k7.double().mean()), float(fx["adv_mean"]), rtol=0, atol=1e-5 * float(fx["adv_std"]))
    np.testing.assert_allclose(float(adv_k7.double().std()), float(fx["adv_std"]), rtol=1e-5)
    with torch.no_grad():
        pi1, _ = actor(t(obs, torch.float32), float(fx["eps"]))
    np.testing.assert_allclose(pi1.cpu().numpy(), fx["pi1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(actor.fc3.bias.detach().cpu().numpy(), fx["actor_fc3_b"], rtol=1e-4, atol=1e-6)
    # the diagnostics the reference logs for this step (coma_mission.py:270-424)
    from ippmarl import metrics
    from test_learning_cpu import ACTOR_TAGS, CRITIC_TAGS
    cm = metrics.critic_metrics([dict(cl.last, discounted=torch.zeros(60, device=dev))], critic)
    am = metrics.actor_metrics([al.last], actor, [pi1])
    assert list(cm) == CRITIC_TAGS and list(am) == ACTOR_TAGS
    np.testing.assert_allclose([cm[k] for k in CRITIC_TAGS], fx["critic_metrics"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose([am[k] for k in ACTOR_TAGS[:7]], fx["actor_metrics"][:7], rtol=1e-3, atol=1e-6)
    # L1 norms of the actor's gradients: sums of strongly cancelling terms (advantages of both signs), so MIOpen's
    # float32 summation order shows up at the per-cent level while the Adam update above still agrees to 1e-4
    np.testing.assert_allclose([am[k] for k in ACTOR_TAGS[7:]], fx["actor_metrics"][7:], rtol=3e-2, atol=1e-6)


# c2 x 64 envs: BASELINE config 3 at its real grid, and c2 x 1024: config 3 at its real size (the round bench.py times:
# 61 440 transitions per wave); c4 x 8 envs: config 4's per-GPU shape (8 UAVs, 512 x 512)
@pytest.mark.parametrize("name,n_envs", [("small", 6), ("c2", 64), ("c4", 8), ("c2", 1024)])
def test_trainer_round_and_td_chains(name, n_envs):
    from ippmarl.trainer import COMATrainer
    params = make_params(name)
    torch.manual_seed(0)
    tr = COMATrainer(params, n_envs=n_envs, waves_per_update=2, first_episode=3)
    s1 = tr.rollout("train")
    s2 = tr.rollout("train")
    assert s1["faults"] == 0 and np.isfinite(s1["episode_return"]) and np.isfinite(s2["episode_return"])
    W, T, E, N = 2, tr.T, tr.E, tr.N
    # TD(lambda) through the kernel == the oracle's literal restatement on each (env, agent) chain
    td, dr = tr.td_targets()
    td = td.view(W, T, E, N).cpu().numpy()
    rew = tr.buf_reward[:W].cpu().numpy()
    g, lam = params["networks"]["gamma"], params["networks"]["lambda"]
    dones = [t == T - 1 for _ in range(W) for t in range(T)]
    for e in ((0, 3, 5) if E < 100 else (0, 3, 5, E // 2, E // 2 + 1, E - 7, E - 2, E - 1)):   # 8 sampled envs at full size
        for i in (0, N - 1):
            # Q of this chain's own W * T states, evaluated on their own (not sliced out of one forward pass over the whole
            # buffer: at 1024 envs that is 131 072 samples whose conv1 activations pass 4 GB, where the library's 32-bit
            # buffer offsets wrap -- the reason COMATrainer.td_targets feeds the critic 16 384 samples at a time)
            with torch.no_grad():
                q, _ = tr.frozen_target(tr.buf_state[:W, :, e, i].reshape(W * T, 11, 11, 12).contiguous())
            q_sel = q.gather(1, tr.buf_action[:W, :, e, i].reshape(-1, 1).long()).view(-1).cpu().numpy()
            want, _ = O.td_lambda_targets(rew[:, :, e].reshape(-1), dones, q_sel, g, lam)
            np.testing.assert_allclose(td[:, :, e, i].reshape(-1), want, rtol=1e-5, atol=2e-6)
    assert td[1, 0].max() == 0.0 and td[1, 0].min() == 0.0  # first step of a later episode in the chain
    before = [p.detach().clone() for p in tr.actor.parameters()]
    stats = tr.update()
    assert stats["transitions"] == W * T * E * N and stats["adam_steps"] == 50
    assert np.isfinite(stats["critic_loss"]) and np.isfinite(stats["actor_loss"])
    changed = [not torch.equal(a, b.detach()) for a, b in zip(before, tr.actor.parameters())]
    names = [n for n, _ in tr.actor.named_parameters()]
    assert all(c for c, n in zip(changed, names) if not n.startswith("fc2"))
    assert not any(c for c, n in zip(changed, names) if n.startswith("fc2"))
    # eval rollout (argmax policy) does not touch the buffer
    tr.rollout("eval")
    assert tr.filled == 0


def test_coma_mission_cadence_and_scalar_names(tmp_path):
    """COMAMission.execute: updates, the reference's TensorBoard tags per training step, eval rounds, best-model pickle."""
    from ippmarl.checkpoint import load_reference_actor
    from ippmarl.missions.mission_factories import MissionFactory
    from ippmarl.utils.writers import ScalarLog
    from test_learning_cpu import ACTOR_TAGS, CRITIC_TAGS
    params = make_params("small", experiment__missions__n_episodes=4, experiment__missions__patience=2)
    factory = MissionFactory(params, log_dir=str(tmp_path), eval_every=2, eval_episodes=7)
    assert isinstance(factory.writer, ScalarLog)          # no tensorboard in this image
    mission = factory.create_mission()
    assert mission.n_envs == 5                            # 300 transitions / (15 steps x 4 UAVs), the reference's round
    best = mission.execute()
    log = factory.writer.scalars
    returns = [f"{m}{k}/{s}" for m in ("train", "eval") for k in ("Return/Episode", "Rewards/Episode", "Return/Relative(used)/Episode")
               for s in ("mean", "std", "max", "min")]
    assert set(log) == set(returns + CRITIC_TAGS + ACTOR_TAGS)
    assert [s for s, _ in log["Critic/Loss"]] == [1, 2, 3, 4] and [s for s, _ in log["evalReturn/Episode/mean"]] == [2, 4]
    assert all(np.isfinite(v) for series in log.values() for _, v in series)
    assert mission.environment_step_idx == 4 * 300 and sum(mission.last_counts["actions"]) == 2 * 5 * 15 * 4
    assert np.isfinite(best) and best == mission.max_mean_episode_return
    actor = load_reference_actor(str(tmp_path / "best_model.pth"), params)   # whole-module pickle, reference style
    assert sum(p.numel() for p in actor.parameters()) == 2275846


def test_sampling_policy_matches_oracle():
    """POLICY_SAMPLE: the device's inverse-CDF draw over probs*mask == the oracle's float32 restatement."""
    from ippmarl.vec_env import VecEnv, POLICY_SAMPLE, POLICY_ARGMAX
    params = make_params("small")
    seed = 77
    env = VecEnv(params, 32, philox_seed=seed)
    eps = np.arange(50, 82)
    env.reset(eps)
    d = env.d
    rng = np.random.RandomState(1)
    for t in range(3):
        env.build_observations(t, features=False)
        probs = rng.dirichlet(np.ones(d.n_actions), size=(env.E, d.n_agents)).astype(np.float32)
        pre = env.pos.cpu().numpy().copy()
        env.steps(t, policy=POLICY_SAMPLE, probs=torch.tensor(probs, device=env.device), features=False)
        act, mask = env.action.cpu().numpy(), env.mask.cpu().numpy()
        for e in range(env.E):
            for i in range(d.n_agents):
                want = O.sample_masked_action(O.philox_action_word(seed, int(eps[e]), i, t), probs[e, i] * mask[e, i])
                assert act[e, i] == want, (t, e, i)
    env.build_observations(3, features=False)
    probs = rng.dirichlet(np.ones(d.n_actions), size=(env.E, d.n_agents)).astype(np.float32)
    env.steps(3, policy=POLICY_ARGMAX, probs=torch.tensor(probs, device=env.device), features=False)
    act, mask = env.action.cpu().numpy(), env.mask.cpu().numpy()
    assert np.array_equal(act, np.argmax(probs * mask, axis=-1))


def test_trainer_round_with_mixed_team_sizes():
    """COMATrainer(team_sizes=...): a learned-policy rollout and a COMA update over a batch whose envs fly 1 .. 4 of 4 UAVs: the
    agents that do not fly stay where they started, get all-zero observations, publish nothing, and none of their transitions
    enters a minibatch; TD(lambda) chains of flying agents against the oracle's restatement."""
    from ippmarl.trainer import COMATrainer
    params = make_params("small")
    teams = [1, 2, 3, 4, 4, 2]
    torch.manual_seed(5)
    tr = COMATrainer(params, n_envs=len(teams), first_episode=3, team_sizes=teams)
    E, N, T = tr.E, tr.N, tr.T
    tr.keep_rollout_log = True
    start = None
    stats = tr.rollout("train")
    assert stats["faults"] == 0
    flying = np.array([[i < n_e for i in range(N)] for n_e in teams])
    obs = tr.buf_obs[0].cpu().numpy()                    # [T, E, N, 11, 11, 7]
    assert not obs[:, ~flying].any()                     # all-zero observations for agents that do not fly
    assert np.allclose(obs[:, flying][..., 1].reshape(T, -1, 121).min(axis=2), obs[:, flying][..., 1].reshape(T, -1, 121).max(axis=2))
    want_id = np.concatenate([[(i + 1) / n_e for i in range(n_e)] for n_e in teams])
    np.testing.assert_allclose(obs[0][flying][:, 0, 0, 1], want_id, rtol=1e-6)     # agent-id plane (i + 1) / team size
    comm = tr.env.comm.cpu().numpy()
    for e, n_e in enumerate(teams):
        assert not comm[e, n_e:].any() and not comm[e, :, n_e:].any()
    idx = tr.valid_transitions(1).cpu().numpy()
    assert len(idx) == sum(teams) * T
    agent_of = idx % N
    env_of = (idx // N) % E
    assert all(agent_of[k] < teams[env_of[k]] for k in range(len(idx)))
    td, _ = tr.td_targets()
    td = td.view(1, T, E, N).cpu().numpy()
    rew = tr.buf_reward[:1].cpu().numpy()
    g, lam = params["networks"]["gamma"], params["networks"]["lambda"]
    dones = np.zeros(T, dtype=bool)
    dones[T - 1] = True
    for e, i in ((0, 0), (1, 1), (2, 2), (5, 1)):
        with torch.no_grad():
            q, _ = tr.frozen_target(tr.buf_state[0, :, e, i].reshape(T, 11, 11, 12).contiguous())
        q_sel = q.gather(1, tr.buf_action[0, :, e, i].reshape(-1, 1).long()).view(-1).cpu().numpy()
        want, _ = O.td_lambda_targets(rew[0, :, e], dones, q_sel, g, lam)
        np.testing.assert_allclose(td[0, :, e, i], want, rtol=1e-5, atol=2e-6)
    before = [p.detach().clone() for p in tr.actor.parameters()]
    out = tr.update()
    assert out["transitions"] == sum(teams) * T and out["adam_steps"] == 50
    assert np.isfinite(out["critic_loss"]) and np.isfinite(out["actor_loss"])
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, tr.actor.parameters()))
    with pytest.raises(Exception, match="one team size"):
        COMATrainer(params, n_envs=2, graphs=True, team_sizes=[1, 2]).capture_graphs()
    del start


def test_evaluation_metrics_match_oracle():
    """COMATrainer.map_metrics / evaluate: target entropy and F1 of the global maps against the oracle's restatement."""
    from ippmarl.trainer import COMATrainer
    params = make_params("small")
    d = O.Derived(params)
    torch.manual_seed(1)
    tr = COMATrainer(params, n_envs=5, first_episode=9)
    out = tr.evaluate(waves=1)
    assert len(out["target_entropy"]) == tr.T + 1 and out["target_entropy"][0] == pytest.approx(1.0)
    assert out["f1"][0] == 0.0 and np.isfinite(out["episode_return"])
    assert out["target_entropy"][-1] < out["target_entropy"][0]
    ent, f1 = tr.map_metrics()
    glob = tr.env.posterior_global().cpu().numpy()
    truth = tr.env.truth_map.numpy().astype(np.float64)
    for e in range(tr.E):
        np.testing.assert_allclose(float(ent[e]), O.target_entropy(d, glob[e].astype(np.float64), truth[e]), rtol=1e-5)
    # cells whose observations cancel exactly sit at log-odds +-1e-7: which side of p = 0.5 they fall on is rounding noise (in
    # the reference as well), so F1 is compared through its attainable range
    from ippmarl import _ffi
    env = tr.env

    def counts(thr):
        c = torch.zeros(tr.E, 3, dtype=torch.int64, device=env.device)
        env.ctx.call("ippm_f1_counts", env._p(env.glob), env._p(env.truth), 1, thr, _ffi.ptr(c), tr.E, env.stream)
        return c.cpu().numpy().astype(np.float64)

    strict, loose = counts(1e-5), counts(-1e-5)
    for e in range(tr.E):
        worst = 2 * strict[e, 0] / (2 * strict[e, 0] + loose[e, 1] + strict[e, 2])
        best = 2 * loose[e, 0] / (2 * loose[e, 0] + strict[e, 1] + loose[e, 2])
        want = O.f1_target(glob[e], truth[e])
        assert worst - 1e-9 <= want <= best + 1e-9 and worst - 1e-9 <= float(f1[e]) <= best + 1e-9
        assert np.array_equal(strict[e, [0, 2]].sum(), truth[e].sum())          # tp + fn = number of target cells
        # the decidable cells AS INTEGERS: the counts under both thresholds equal those of the exported probabilities (every cell
        # but the exactly-cancelled ones is at least one measurement's log-odds away from 0)
        assert tuple(int(v) for v in strict[e]) == O.f1_counts(glob[e], truth[e], 1e-5), e
        assert tuple(int(v) for v in loose[e]) == O.f1_counts(glob[e], truth[e], -1e-5), e


def test_evaluate_scores_the_freshly_sensed_measurements():
    """COMATrainer.evaluate() against the oracle's restatement of the coma_test loop (coma_test.py:98-196): the map scored at
    index t+1 holds the measurements taken right AFTER the move of step t (the env's own global fusion lags by one step).
    A stub actor with fixed preferences makes the greedy choice a function of the masks alone, so the oracle can follow."""
    from ippmarl.trainer import COMATrainer
    params = make_params("small", experiment__missions__n_agents=3)
    d = O.Derived(params)
    seed, E, first = 77, 3, 21
    tr = COMATrainer(params, n_envs=E, philox_seed=seed, first_episode=first)
    prefs = torch.tensor([0.05, 0.3, 0.1, 0.25, 0.2, 0.1])

    class Stub(torch.nn.Module):
        def forward(self, obs, eps):
            return prefs.to(obs.device).expand(obs.shape[0], -1).contiguous(), None

    tr.actor = Stub()
    dev_counts = []
    out = tr.evaluate(waves=1, counts_log=dev_counts)
    ents, f1s, want_counts = [], [], []
    for ep_no in range(first, first + E):
        holder, seen = {}, {}

        def correctness(i, s, shape, ep_no=ep_no, holder=holder):
            pos = holder["ep"].agents[i]["position"]
            _, fc = O.project_field_of_view(d, pos)
            return O.philox_correctness(seed, ep_no, i, s, fc, d.gy, O.noise_of_altitude(pos[2]))

        ep = O.OracleEpisode(params, ep_no, correctness, lambda i, t, m, o: int(np.argmax(prefs.numpy() * m)),
                             comm_draw=lambda i, j, t, ep_no=ep_no: O.philox_comm_draw(seed, ep_no, i, j, t), build_features=False,
                             exact=True)
        holder["ep"] = ep
        real_sense = ep._sense

        def sense(i, s, ep=ep, seen=seen, real_sense=real_sense):
            real_sense(i, s)
            seen[(i, s)] = ep.agents[i]["map2communicate"]

        ep._sense = sense
        g = O.init_prior_map(ep.d)
        with O.record_f1_counts() as counts:
            ent, f1 = [O.target_entropy(ep.d, g.copy(), ep.truth)], [O.f1_target(g, ep.truth)]
            for t in range(d.budget + 1):
                ep.step(t)
                if t == 0:
                    g = O.fuse_map(ep.d, g, {i: dict(map2communicate=seen[(i, 0)]) for i in range(d.n_agents)}, None, "global")
                g = O.fuse_map(ep.d, g, [seen[(i, t + 1)] for i in range(d.n_agents)], None, "global")
                ent.append(O.target_entropy(ep.d, g.copy(), ep.truth))
                f1.append(O.f1_target(g, ep.truth))
        ents.append(ent)
        f1s.append(f1)
        want_counts.append(counts)
    np.testing.assert_allclose(out["target_entropy"], np.mean(ents, axis=0), rtol=1e-5)
    # F1 thresholds at p > 0.5: exactly-cancelled cells are rounding noise on either side
    np.testing.assert_allclose(out["f1"], np.mean(f1s, axis=0), atol=0.05)
    # ... and the decidable cells as integers: (tp, fp, fn) of every scored map of every env under the thresholds +-1e-5
    assert len(dev_counts) == d.budget + 2
    for k, (strict, lax) in enumerate(dev_counts):
        for e in range(E):
            assert tuple(strict[e].tolist()) == want_counts[e][k][0], (k, e, "log-odds > +1e-5")
            assert tuple(lax[e].tolist()) == want_counts[e][k][1], (k, e, "log-odds > -1e-5")
    assert abs(out["f1"][-1] - np.mean(f1s, axis=0)[-1]) < 0.05 and out["f1"][1] > 0


def test_interleaved_update_equals_reference_order():
    """COMATrainer.update() runs actor(b) right after critic(b) so that one all-reduce carries actor(b) + critic(b+1); the
    reference runs all critic minibatches, then all actor minibatches (critic/learner.py:58-105, actor/learner.py:36-101).
    Same minibatches, same numbers: neither net's step feeds the other's."""
    from ippmarl.trainer import COMATrainer
    params = make_params("small")

    def fresh():
        torch.manual_seed(3)
        tr = COMATrainer(params, n_envs=4, first_episode=5)
        tr.rollout("train")
        return tr

    a, b = fresh(), fresh()
    torch.manual_seed(11)
    a.update()
    torch.manual_seed(11)
    W, T, E, N = b.filled, b.T, b.E, b.N
    n = W * T * E * N
    td, _ = b.td_targets()
    obs, states = b.buf_obs[:W].reshape(n, 11, 11, 7), b.buf_state[:W].reshape(n, 11, 11, 12)
    actions, masks = b.buf_action[:W].reshape(n), b.buf_mask[:W].reshape(n, b.A)
    bs = n // b.batch_number
    for data_pass in range(b.data_passes):
        perm = torch.randperm(n, device=b.device)
        b.critic_learner.update_target_network(b.train_step, data_pass)
        q_new = []
        for k in range(b.batch_number):
            idx = perm[k * bs:(k + 1) * bs]
            q_new.append(b.critic_learner.step(states[idx], actions[idx], td[idx])[1])
        for k in range(b.batch_number):
            idx = perm[k * bs:(k + 1) * bs]
            b.actor_learner.step(obs[idx], actions[idx], masks[idx], q_new[k], b.eps)
        if data_pass == 0:
            b.train_step += 1
    # MIOpen's weight-gradient kernels split K with atomics, so two runs of the SAME schedule differ in the last bits and Adam
    # turns a near-zero gradient's sign into +-lr: nearly all weights agree to 1e-6, stragglers stay within a few lr (1e-4)
    for pa, pb in zip(list(a.actor.parameters()) + list(a.critic.parameters()), list(b.actor.parameters()) + list(b.critic.parameters())):
        assert float(((pa - pb).abs() <= 1e-6).float().mean()) >= 0.98
        torch.testing.assert_close(pa, pb, rtol=0, atol=5e-4)
    assert a.reducer.flat is not None and a.actor.conv1.weight.grad.data_ptr() >= a.reducer.flat.data_ptr()


def test_conv2_input_gradient_through_gemm_and_col2im_kernel():
    """networks._ConvDataGradAsGemm on the GPU (hipBLASLt GEMM + ippm_col2im_nhwc) against the library's own convolution
    backward: same outputs, same gradients for every parameter and for the input."""
    from ippmarl import networks as N
    params = make_params("c2")
    torch.manual_seed(4)
    net = N.CriticNetwork(params).cuda()
    x = torch.rand(96, 11, 11, 12, device="cuda", requires_grad=True)

    def run(flag):
        old, N.CONV2_BWD_DATA_AS_GEMM = N.CONV2_BWD_DATA_AS_GEMM, flag
        try:
            net.zero_grad()
            x.grad = None
            q, _ = net(x)
            (q * q).sum().backward()
            return q.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        finally:
            N.CONV2_BWD_DATA_AS_GEMM = old

    q0, gx0, g0 = run(False)
    q1, gx1, g1 = run(True)
    torch.testing.assert_close(q1, q0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gx1, gx0, rtol=2e-4, atol=1e-6 * float(gx0.abs().max()))
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], rtol=2e-4, atol=2e-5 * float(g0[n].abs().max()), msg=n)
    # the kernel alone against the definition (every column element lands in exactly one output sum)
    from ippmarl import _ffi
    B, Ho, Wo, K, C = 3, 4, 4, 4, 8
    cols = torch.rand(B * Ho * Wo, K * K * C, device="cuda")
    out = torch.empty(B, Ho + K - 1, Wo + K - 1, C, device="cuda")
    _ffi.check(_ffi.load_library().ippm_col2im_nhwc(_ffi.ptr(cols), _ffi.ptr(out), B, Ho, Wo, K, C,
                                                    torch.cuda.current_stream().cuda_stream), "ippm_col2im_nhwc")
    want = torch.zeros_like(out)
    cv = cols.view(B, Ho, Wo, K, K, C)
    for ky in range(K):
        for kx in range(K):
            want[:, ky:ky + Ho, kx:kx + Wo] += cv[:, :, :, ky, kx]
    torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-6)


def test_fused_bias_relu_equals_separate_passes():
    """networks._BiasReLU (one in-place pass forward, one pass backward incl. the bias gradient) against PyTorch's
    conv-with-bias + ReLU: the kernels alone on ragged row counts, then both convnets end to end with the fusion on / off."""
    from ippmarl import _ffi
    from ippmarl import networks as N
    lib = _ffi.load_library()
    stream = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(7)
    for rows, C in ((1, 256), (513, 256), (4096 * 49 + 3, 256), (1000, 64), (77, 8)):
        x = torch.randn(rows, C, device="cuda")
        b = torch.randn(C, device="cuda")
        want = torch.relu(x + b)
        y = x.clone()
        _ffi.check(lib.ippm_bias_relu_nhwc(_ffi.ptr(y), _ffi.ptr(b), rows, C, stream), "ippm_bias_relu_nhwc")
        assert torch.equal(y, want), (rows, C)
        gy = torch.randn(rows, C, device="cuda")
        gx = torch.empty_like(gy)
        gb = torch.zeros(C, device="cuda")
        _ffi.check(lib.ippm_bias_relu_backward_nhwc(_ffi.ptr(gy), _ffi.ptr(y), _ffi.ptr(gx), _ffi.ptr(gb), rows, C, stream),
                   "ippm_bias_relu_backward_nhwc")
        want_gx = gy * (want > 0)
        assert torch.equal(gx, want_gx), (rows, C)
        # float32 sums in a different order: 1e-5 of the column's absolute mass
        torch.testing.assert_close(gb, want_gx.sum(0), rtol=0, atol=1e-5 * float(want_gx.abs().sum(0).max()) + 1e-6)
    with pytest.raises(_ffi.IppmError):
        _ffi.check(lib.ippm_bias_relu_nhwc(_ffi.ptr(y), _ffi.ptr(b), 4, 6, stream), "ippm_bias_relu_nhwc")   # 6 channels

    params = make_params("c2")
    for cls, planes in ((N.ActorNetwork, 7), (N.CriticNetwork, 12)):
        net = cls(params).cuda()
        x = torch.rand(160, 11, 11, planes, device="cuda")

        def run(flag):
            old, N.FUSED_BIAS_RELU = N.FUSED_BIAS_RELU, flag
            try:
                net.zero_grad()
                out = net(x, 0.1)[0] if cls is N.ActorNetwork else net(x)[0]
                (out * out).sum().backward()
                with torch.no_grad():
                    out_ng = net(x, 0.1)[0] if cls is N.ActorNetwork else net(x)[0]
                return out.detach().clone(), out_ng.clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
            finally:
                N.FUSED_BIAS_RELU = old

        o0, n0, g0 = run(False)
        o1, n1, g1 = run(True)
        torch.testing.assert_close(o1, o0, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(n1, n0, rtol=1e-5, atol=1e-6)
        assert set(g0) == set(g1)
        for n in g0:
            torch.testing.assert_close(g1[n], g0[n], rtol=2e-4, atol=2e-5 * float(g0[n].abs().max()), msg=n)


def test_forward_pass_over_a_whole_buffer_is_sliced():
    """131 072 critic states in one call (the whole buffer of a 1024-env round): conv1's activations would pass 4 GB, where the
    library's 32-bit buffer offsets wrap and samples 85 598.. land on samples 0..; the trunk slices such batches, so the
    result equals the per-slice evaluation everywhere."""
    from ippmarl.networks import CriticNetwork
    torch.manual_seed(3)
    net = CriticNetwork(make_params("c2")).cuda().eval()
    x = torch.rand(131072, 11, 11, 12, device="cuda")
    with torch.no_grad():
        whole, _ = net(x)
        for lo in (0, 40000, 85598, 120000):
            part, _ = net(x[lo:lo + 4096])
            torch.testing.assert_close(whole[lo:lo + 4096], part, rtol=1e-5, atol=1e-6)


def test_recorded_round_equals_eager_round():
    """COMATrainer.capture_graphs(): a round replayed from hipGraphs (16 rollout-step graphs + one graph of TD targets and the
    25 + 25 Adam steps) is the round run launch by launch: the same actions over bit-identical maps; the weights as closely
    as two launch-by-launch runs agree with each other (the gradient kernels sum with float atomics, so even those differ in the
    last bits, and Adam's normalised step turns a last-bit change of a near-zero gradient into a visible one)."""
    from ippmarl.trainer import COMATrainer
    params = make_params("small")

    def trainer():
        torch.manual_seed(11)
        return COMATrainer(params, n_envs=5, first_episode=3, graphs=True)

    eager, eager2, rec = trainer(), trainer(), trainer()
    trio = (eager, eager2, rec)
    for tr in trio:   # first round launch by launch on all (kernel selection, allocator)
        torch.manual_seed(12)
        tr.rollout("train")
        tr.update()
    # The first round leaves the three trainers different: in the last bits at least (float atomics in the gradient kernels), and by
    # whole steps when a sampled action of that round sat on an inverse-CDF boundary in one of them (the very first forward passes
    # of the process run while the convolution library is still choosing its kernels).  The rounds compared below therefore start
    # from IDENTICAL training state -- weights AND both Adam instances' moments and step counters, copied in place (the graphs hold
    # the tensors' addresses) -- so that "the same round" is a statement about the replay and not about luck.  (Until round 5 only
    # the weights were copied: a first round that had diverged left other Adam moments behind, the later rounds then differed by
    # 20 - 40 % of a step between the two EAGER trainers as well, and the comparison failed in about one run in six.)
    def sync_state():
        with torch.no_grad():
            for tr in (eager2, rec):
                for net in ("actor", "critic"):
                    for p_dst, p_src in zip(getattr(tr, net).parameters(), getattr(eager, net).parameters()):
                        p_dst.copy_(p_src)
                for learner in ("actor_learner", "critic_learner"):
                    src_opt, dst_opt = getattr(eager, learner).optimizer, getattr(tr, learner).optimizer
                    for g_src, g_dst in zip(src_opt.param_groups, dst_opt.param_groups):
                        for p_src, p_dst in zip(g_src["params"], g_dst["params"]):
                            st_src, st_dst = src_opt.state.get(p_src, {}), dst_opt.state.get(p_dst, {})
                            assert set(st_src) == set(st_dst)
                            for k, v in st_src.items():
                                st_dst[k].copy_(v)
                for p_dst, p_src in zip(tr.critic_learner.target_critic.parameters(), eager.critic_learner.target_critic.parameters()):
                    p_dst.copy_(p_src)

    sync_state()
    rec.capture_graphs()

    def flat(net):
        return torch.cat([p.detach().reshape(-1) for p in net.parameters()])

    for rnd in range(2):
        if rnd:
            sync_state()     # (the second round too: after the first the weights differ in their last bits again)
        before = flat(eager.actor).clone(), flat(eager.critic).clone()
        bufs = []
        for tr in trio:
            torch.manual_seed(20 + rnd)   # the minibatch permutations
            s = tr.rollout("train")
            assert s["faults"] == 0
            bufs.append({k: getattr(tr, k).clone() for k in ("buf_obs", "buf_state", "buf_action", "buf_mask", "buf_reward")})
            stats = tr.update()
            assert stats["adam_steps"] == 50 and np.isfinite(stats["critic_loss"]) and np.isfinite(stats["actor_loss"])
        # the replayed rollout took the same actions over the same maps; network inputs and rewards agree to the summation
        # order of the float64 atomics behind the tracked area sums / reward sums
        for k in ("buf_action", "buf_mask"):
            assert torch.equal(bufs[0][k], bufs[2][k]), k
        for k in ("buf_obs", "buf_state", "buf_reward"):
            torch.testing.assert_close(bufs[0][k], bufs[2][k], rtol=1e-5, atol=2e-6, msg=k)
        assert torch.equal(eager.env.local, rec.env.local) and torch.equal(eager.env.glob, rec.env.glob)
        for net, b in (("actor", before[0]), ("critic", before[1])):
            e1, e2, r = flat(getattr(eager, net)), flat(getattr(eager2, net)), flat(getattr(rec, net))
            moved = float((e1 - b).norm())
            assert moved > 0
            d_ee, d_er = float((e1 - e2).norm()), float((e1 - r).norm())
            # (a wrong permutation, a missed step or a stale buffer would put d_er at the size of `moved` itself)
            assert d_er <= max(10.0 * d_ee, 5e-2 * moved), (rnd, net, d_er, d_ee, moved)
    assert eager.train_step == rec.train_step == 3


def test_coma_loop_learns():
    """End to end: 8 COMA updates (1024 episodes each: rollout with the epsilon-mixed actor, TD(lambda) targets, 25 + 25 Adam steps,
    the reference's hyper-parameters and its frozen target critic) turn the untrained actor into a greedy policy that beats
    the uniform random walk on 1024 FIXED evaluation episodes it never trained on (same truth, start cells and sensor noise for
    every policy: all streams are keyed by the episode number).  The reference's own yardstick: coma_test.py:84-97,177-196 against
    random_baseline.py:91-96; cadence missions/coma_mission.py:48-172.
    Measured (profiles/r06/learning_curve*.json): random walk -2.22, untrained actor -3.99 / -4.72 (seeds 0 / 1), after 8 updates
    -1.00 / -0.95, after 200 updates +5.68 (the greedy information-gain planner: +2.89).  Margins asserted: +0.6 over the random
    walk (a third of the measured 1.2), +2.0 over the untrained actor (measured 3.0 / 3.8).  About 15 s."""
    from ippmarl.params import grid256_params
    from ippmarl.trainer import COMATrainer
    params = grid256_params(experiment__missions__n_agents=4)
    torch.manual_seed(0)
    tr = COMATrainer(params, 1024)
    eval_ids = torch.arange(100_000_001, 100_000_001 + 1024, dtype=torch.int64)
    random_walk = tr.returns_on(eval_ids, "random")
    untrained = tr.returns_on(eval_ids, "actor")
    assert random_walk["faults"] == 0 and untrained["faults"] == 0
    wave0 = tr.wave
    for _ in range(8):
        stats = tr.rollout("train")
        stats.update(tr.update())
        assert stats["faults"] == 0 and np.isfinite(stats["critic_loss"]) and np.isfinite(stats["actor_loss"])
    assert tr.wave == wave0 + 8            # (evaluation on fixed episodes does not advance the training waves)
    trained = tr.returns_on(eval_ids, "actor")
    again = tr.returns_on(eval_ids, "actor")
    # the evaluation is a function of the weights and the episodes only (returns to the summation order of the reward atomics)
    assert abs(again["episode_return"] - trained["episode_return"]) < 1e-4
    assert trained["episode_return"] > random_walk["episode_return"] + 0.6, (trained, random_walk)
    assert trained["episode_return"] > untrained["episode_return"] + 2.0, (trained, untrained)
    assert trained["final_f1"] > untrained["final_f1"]

