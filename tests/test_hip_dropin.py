"""GPU tests of the drop-in object surface (COMAWrapper / Agent / Mapping / AgentActionSpace / ... with the reference's
signatures), driven the way the reference's own scripts drive them and checked against the reference's recorded outputs."""
import numpy as np
import pytest

import ipp_oracle as O
from configs import make_params
from conftest import REFERENCE_QUANTISATION_CELLS as RQ, assert_posteriors, local_subset, unpack_correctness
from test_oracle_golden import EPISODES

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.mark.parametrize("tag", ["episode_c2_e1", "episode_small5_e3", "episode_default_e2", "episode_c4_e2"])
def test_episode_generator_replays_reference_episode(golden, tag, monkeypatch):
    """EpisodeGenerator.execute(...) exactly as missions/coma_mission.py calls it, with the recorded randomness."""
    from ippmarl.batch_memory import BatchMemory
    from ippmarl.coma_wrapper import COMAWrapper, ReplayHooks
    from ippmarl.mapping.grid_maps import GridMap
    from ippmarl.missions.episode_generator import EpisodeGenerator
    from ippmarl.sensors import Sensor
    from ippmarl.sensors.models import SensorModel
    fx = golden(tag)
    params = make_params(EPISODES[tag]["name"], **EPISODES[tag]["over"])
    n = params["experiment"]["missions"]["n_agents"]
    T = params["experiment"]["constraints"]["budget"] + 1
    corr = unpack_correctness(fx)
    draws = iter(fx["comm_draws"])
    monkeypatch.setattr(np.random, "random_sample", lambda *a, **k: next(draws))
    wrapper = COMAWrapper(params, None)
    wrapper.replay = ReplayHooks(correctness=lambda agent_id, stage: corr[stage * n + agent_id],
                                 action=lambda agent_id, t: int(fx["actions"][t, agent_id]))
    memory = BatchMemory(params, wrapper)
    grid_map = GridMap(params)
    generator = EpisodeGenerator(params, None, grid_map, Sensor(SensorModel(), grid_map))
    mode = "eval" if tag == "episode_small5_e3" else "train"
    (episode_return, episode_rewards, absolute_return, simulated_map, memory, agent_positions, t_last, eps, agent_actions,
     agent_altitudes) = generator.execute(int(fx["episode"]), memory, wrapper, mode)
    assert np.array_equal(simulated_map.astype(np.uint8), fx["truth"])
    assert np.array_equal(np.array(agent_positions), fx["positions"])           # [T+1, n, 3], bit-exact
    assert np.array_equal(np.array(agent_actions), fx["actions"])
    np.testing.assert_allclose(episode_rewards, fx["episode_rewards"], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(episode_return, fx["episode_return"], rtol=RTOL)
    np.testing.assert_allclose(absolute_return, fx["abs_return"], rtol=RTOL)
    np.testing.assert_allclose(eps, fx["eps"], rtol=1e-12)
    assert t_last == T - 1 and memory.size() == T * n
    for t in range(T):
        for a in range(n):
            tr = memory.transitions[a][t]
            assert np.array_equal(tr.mask.cpu().numpy(), fx["masks"][t, a])
            np.testing.assert_allclose(tr.observation.cpu().numpy(), fx["obs"][t, a], rtol=RTOL, atol=2e-6)
            np.testing.assert_allclose(tr.state.cpu().numpy(), fx["state"][t, a], rtol=RTOL, atol=2e-6)
            assert tr.done == bool(fx["done"][t, a]) and abs(tr.reward - fx["rewards"][t, a]) <= 1e-5 * abs(fx["rewards"][t, a]) + 1e-6
    assert_posteriors(local_subset(fx, np.array([ag.local_map for ag in generator.agents])), fx["final_local"], strict=False, msg="final local", allow=RQ.get((tag, "final_local"), []))


def test_action_space_matches_reference_tables(golden):
    from ippmarl.agent.action_space import AgentActionSpace
    from ippmarl.agent.state_space import AgentStateSpace
    fx = golden("masks")
    for A in (4, 6, 9, 27):
        over = dict(experiment__constraints__num_actions=A)
        if A in (4, 9):  # the 2-D sets are only self-consistent with one altitude level (the fixtures were recorded that way)
            over.update(experiment__constraints__min_altitude=15, experiment__constraints__max_altitude=15)
        params = make_params("default", **over)
        asp, ss = AgentActionSpace(params), AgentStateSpace(params)
        for pos, want in list(zip(fx[f"a{A}_pos"], fx[f"a{A}_mask"]))[::7]:
            flat, _ = asp.get_action_mask(pos)
            assert np.array_equal(flat, want), (A, pos)
        base = np.array([25, 25, 15 if A in (4, 9) else 10])
        assert np.array_equal(np.array([asp.action_to_position(base, a) for a in range(A)]), fx[f"a{A}_moves"])
        for pos, others, k, m_in, m_out in list(zip(fx[f"a{A}_col_pos"], fx[f"a{A}_col_others"], fx[f"a{A}_col_n"],
                                                    fx[f"a{A}_col_in"], fx[f"a{A}_col_out"]))[:200]:
            got = asp.apply_collision_mask(pos, m_in.copy(), [others[j] for j in range(k)], ss)
            assert np.array_equal(got, m_out), (A, pos, others[:k])


def test_camera_and_state_space(golden):
    from ippmarl.agent.state_space import AgentStateSpace
    from ippmarl.mapping.grid_maps import GridMap
    from ippmarl.sensors.cameras import Camera
    from ippmarl.sensors.models.sensor_models import AltitudeSensorModel
    fx = golden("derived_footprints")
    params = make_params("c2")
    gm, ss = GridMap(params), AgentStateSpace(params)
    cam = Camera(params, AltitudeSensorModel(params), gm)
    k = 0
    for x in range(11):
        for y in range(11):
            for z in range(3):
                if k % 11 == 0:
                    full, clip = cam.project_field_of_view(ss.index_to_position([x, y, z]), gm.resolution_x, gm.resolution_y)
                    assert full == list(fx["c2_fp_full"][k]) and clip == list(fx["c2_fp_clip"][k])
                k += 1
    st = golden("start_states")["seed3"]
    assert np.array_equal(np.array([[ss.get_random_agent_state(a, e) for a in range(4)] for e in range(1, 9)]), st[:8, :4])
    assert AltitudeSensorModel(params).get_noise_variance(10) == 0.265


def test_mapping_update_fuse_and_reward_standalone(golden):
    """Mapping.update_grid_map / fuse_map / get_global_reward on plain NumPy arrays, against the oracle."""
    from ippmarl.mapping.grid_maps import GridMap
    from ippmarl.mapping.mappings import Mapping
    from ippmarl.sensors import Sensor
    from ippmarl.sensors.models import SensorModel
    from ippmarl.utils.reward import get_global_reward
    from ippmarl.agent.state_space import AgentStateSpace
    params = make_params("small")
    d = O.Derived(params)
    d.exact = True
    gm = GridMap(params)
    mapping = Mapping(gm, Sensor(SensorModel(), gm), params, 4)
    truth = O.make_truth(d, 4)
    assert np.array_equal(mapping.simulated_map, truth)
    rng = np.random.RandomState(2)
    maps, infos = [], {}
    for i, pos in enumerate([[10, 15, 15], [20, 15, 10], [15, 20, 5]]):
        pos = np.array(pos)
        _, fc = O.project_field_of_view(d, pos)
        corr = (rng.random_sample(O.tile_shape(fc)) > 0.3).astype(np.int64)
        mine = mapping.init_priors()
        ref = O.init_prior_map(d)
        out = mapping.update_grid_map(pos, mine, 0, "train", agent_id=i, correctness=corr)
        want = O.update_grid_map(d, truth, pos, ref, corr)
        assert out[0] is mine and out[2] == want[2]                           # in-place mutation, same rect
        assert_posteriors(mine, ref, strict=True, msg="update_grid_map")
        np.testing.assert_array_equal(np.asarray(out[3]), want[3].astype(np.float32))   # map2communicate
        np.testing.assert_allclose(out[4], want[4], rtol=1e-7)                          # footprint_img
        maps.append(mine)
        infos[i] = {"map2communicate": out[3]}
        infos_ref = infos
    fused = mapping.fuse_map(maps[0], infos, 0, "local")
    want = O.fuse_map(d, maps[0].astype(np.float64), {k: {"map2communicate": np.asarray(v["map2communicate"])} for k, v in infos.items()}, 0, "local")
    assert_posteriors(fused, want, strict=True, msg="fuse_map local")
    glob = mapping.fuse_map(mapping.init_priors(), [np.asarray(infos[k]["map2communicate"]) for k in (0, 1, 2)], None, "global")
    want_g = O.fuse_map(d, O.init_prior_map(d), [np.asarray(infos[k]["map2communicate"]) for k in (0, 1, 2)], None, "global")
    assert_posteriors(glob, want_g, strict=True, msg="fuse_map global (plain arrays)")
    done, rel, ab = get_global_reward(mapping.init_priors(), glob, "COMA", None, truth, AgentStateSpace(params), None, None, 0, 14)
    _, rel_w, ab_w = O.global_reward(d, O.init_prior_map(d), want_g, truth)
    assert done is False
    np.testing.assert_allclose([rel, ab], [rel_w, ab_w], rtol=RTOL, atol=1e-6)


def test_batch_memory_td_targets(golden):
    from ippmarl.batch_memory import BatchMemory
    fx = golden("td_lambda")
    params = make_params("default")
    n, L = fx["rewards"].shape

    class TableCritic(torch.nn.Module):
        def __init__(self, table):
            super().__init__()
            self.table = torch.nn.Parameter(torch.tensor(table), requires_grad=False)

        def forward(self, state):
            return self.table[state.view(-1).long()], None

    rng = np.random.RandomState(0)
    actions = rng.randint(0, 6, size=(n, L))
    table = rng.standard_normal((n * L, 6)).astype(np.float32)
    for a in range(n):
        for i in range(L):
            table[a * L + i, actions[a, i]] = fx["qsel"][a, i]
    bm = BatchMemory(params, None)
    for a in range(n):
        for i in range(L):
            bm.add(a, state=torch.tensor([float(a * L + i)]), action=torch.tensor([actions[a, i]]), reward=float(fx["rewards"][a, i]),
                   done=bool(fx["dones"][a, i]))
    bm.build_td_targets(TableCritic(table))
    td = np.array([[float(bm.transitions[a][i].td_target) for i in range(L)] for a in range(n)])
    dr = np.array([[float(bm.transitions[a][i].discounted_return) for i in range(L)] for a in range(n)])
    np.testing.assert_allclose(td, fx["td"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dr, fx["dr"], rtol=1e-5, atol=2e-6)
    batches = bm.build_batches()
    assert len(batches) == (n * L) // 60 and all(len(b) == 60 for b in batches)


@pytest.mark.parametrize("tag", ["ig_c1_e1", "ig_small3_e4"])
def test_ig_baseline_replays_reference_run(golden, tag, monkeypatch):
    """IG_baseline(params, writer, episode).execute() as IG_baseline.main drives it (BASELINE config 1 and a smaller case),
    with the sensor noise the reference drew: chosen altitudes, candidate gains, target entropy and F1 per step."""
    from ippmarl.IG_baseline import IG_baseline
    from ippmarl.coma_wrapper import ReplayHooks
    from test_oracle_golden import IG_CASES
    fx = golden(tag)
    params = make_params(IG_CASES[tag]["name"], **IG_CASES[tag]["over"])
    n = params["experiment"]["missions"]["n_agents"]
    corr = unpack_correctness(fx)
    draws = iter(fx["comm_draws"])
    monkeypatch.setattr(np.random, "random_sample", lambda *a, **k: next(draws))
    ig = IG_baseline(params, None, int(fx["episode"]))
    ig.replay = ReplayHooks(correctness=lambda agent_id, stage: corr[stage * n + agent_id])
    rel, ab, altitudes, entropies, f1s = ig.execute()
    assert np.array_equal(np.array(altitudes), fx["altitudes"])
    np.testing.assert_allclose(np.array(ig.gains_log), fx["gains"], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(entropies, fx["entropies"], rtol=RTOL)
    # F1 thresholds the map at p > 0.5.  Cells whose observations cancel exactly (47 % of the twice-observed cells at 15 m)
    # sit at 0.5 +- 1e-8 in the reference and are classified by its rounding noise, so the recorded F1 can only be
    # bracketed: every such cell in the wrong class <= reference <= every such cell in the right class.
    assert f1s[0] == fx["f1"][0] == 0.0
    _check_f1_brackets(ig, fx)
    from test_oracle_golden import oracle_ig_run
    _check_f1_counts(ig, oracle_ig_run(fx, tag)[1])
    np.testing.assert_allclose([rel, ab], [fx["relative_return"], fx["absolute_return"]], rtol=1e-9)


def _check_f1_counts(obj, oracle_counts):
    """tp / fp / fn of every evaluation, AS INTEGERS, under the two thresholds that leave the exactly-cancelled cells out
    (log-odds > +1e-5) and take them all in (> -1e-5): every other cell is at least one measurement's log-odds away from 0, so
    these counts do not depend on anyone's rounding -- the device's must equal the oracle's (coma_test.py:177-196,
    utils/utils.py:43-76).  The bracket above is only about the cancelled remainder."""
    assert len(obj.f1_counts_log) == len(oracle_counts)
    for k, (got, want) in enumerate(zip(obj.f1_counts_log, oracle_counts)):
        assert tuple(got[0]) == tuple(want[0]), (k, "log-odds > +1e-5", got[0], want[0])
        assert tuple(got[1]) == tuple(want[1]), (k, "log-odds > -1e-5", got[1], want[1])


def _check_f1_brackets(obj, fx):
    """The recorded F1 lies inside the bracket, and the bracket is no wider than the exactly-cancelled cells allow: moving
    c cells between the classes moves 2tp / (2tp + fp + fn) by at most 2c / (tp + fn) -- a [0, 1] bracket does not pass."""
    assert len(obj.f1_bracket) == len(obj.f1_cancelled) == len(fx["f1"])
    for k, ((lo, hi), share, want) in enumerate(zip(obj.f1_bracket, obj.f1_cancelled, fx["f1"])):
        lo, hi = min(lo, hi), max(lo, hi)
        assert lo - 1e-9 <= want <= hi + 1e-9, (lo, want, hi)
        assert hi - lo <= 2 * share + 1e-9, (lo, hi, share)
        assert hi - lo < 0.2, (k, lo, hi)   # (widest where footprints overlap most: ~0.1 after the start sensing, 0.15 for the sweeps)


def _check_curves(obj, entropies, f1s, fx):
    np.testing.assert_allclose(entropies, fx["entropies"], rtol=RTOL)
    assert f1s[0] == fx["f1"][0] == 0.0
    _check_f1_brackets(obj, fx)   # exactly-cancelled cells: see the IG test above


def test_random_baseline_replays_reference_run(golden):
    """random_baseline.py: one shared map updated by every platform, uniform actions over the boundary mask."""
    from ippmarl.random_baseline import RandomBaseline
    from ippmarl.coma_wrapper import ReplayHooks
    fx = golden("random_small3_e6")
    params = make_params("small", experiment__missions__n_agents=3)
    n, corr = 3, unpack_correctness(fx)
    rb = RandomBaseline(params, None, int(fx["episode"]))
    rb.replay = ReplayHooks(correctness=lambda i, t: corr[t * n + i], action=lambda i, t: int(fx["actions"][(t - 1) * n + i]))
    ret, entropies, f1s = rb.execute()
    assert ret == int(fx["ret"]) == 0
    _check_curves(rb, entropies, f1s, fx)
    from test_oracle_golden import oracle_random_run
    _check_f1_counts(rb, oracle_random_run(fx)[2])
    # without hooks: device noise, host multinomial; the curve must still be a mapping run (entropy falls, F1 rises)
    torch.manual_seed(3)
    _, ent, f1 = RandomBaseline(params, None, 7).execute()
    assert len(ent) == params["experiment"]["constraints"]["budget"] + 2 and ent[0] == 1.0 and ent[-1] < 0.97 and f1[-1] > 0.1
    assert all(b <= a + 1e-9 for a, b in zip(ent, ent[1:]))   # observations only ever remove entropy from the target cells


def test_lawn_mower_replays_reference_run(golden):
    """lawn_mower.py: eight fixed sweeps at the configured altitude update one shared map."""
    from ippmarl.lawn_mower import LawnMower, coverage_paths
    from ippmarl.coma_wrapper import ReplayHooks
    fx = golden("lawnmower_small_e2")
    params = make_params("small", experiment__missions__n_agents=8, experiment__baselines__lawnmower__altitude=10)
    corr = unpack_correctness(fx)
    assert len(corr) == 8 * 15 and all(p.shape == (15, 3) for p in coverage_paths(10))
    lm = LawnMower(params, None, int(fx["episode"]))
    lm.replay = ReplayHooks(correctness=lambda k, idx: corr[idx * 8 + k])
    ret, entropies, f1s = lm.execute()
    assert ret == 0
    _check_curves(lm, entropies, f1s, fx)
    from test_oracle_golden import oracle_lawnmower_run
    _check_f1_counts(lm, oracle_lawnmower_run(fx)[2])
    # the script does not depend on n_agents (the reference needs 8 memory slots; the shared map needs one)
    _, ent2, _ = LawnMower(make_params("small", experiment__baselines__lawnmower__altitude=10), None, 2).execute()
    assert len(ent2) == 16 and ent2[-1] < 0.8


def test_coma_test_replays_reference_run(golden):
    """coma_test.py: greedy deployment of a (here: seeded, untrained) actor; positions, altitudes, entropy and F1 per step."""
    from ippmarl.coma_test import COMATest
    from ippmarl.coma_wrapper import ReplayHooks
    from ippmarl.networks import ActorNetwork
    fx = golden("comatest_small3_e9")
    params = make_params("small", experiment__missions__n_agents=3)
    n, corr = 3, unpack_correctness(fx)
    torch.manual_seed(int(fx["net_seed"]))
    net = ActorNetwork(params)

    def run(force_actions):
        ct = COMATest(params, None, int(fx["episode"]))
        ct.net = net
        ct.replay = ReplayHooks(correctness=lambda i, stage: corr[stage * n + i],
                                action=(lambda i, t: int(fx["actions"][t * n + i])) if force_actions else None)
        return ct, ct.execute("random", int(fx["episode"]))

    ct, (ret, positions, altitudes, entropies, f1s, rel) = run(True)
    assert np.array_equal(np.array(positions), fx["positions"])
    assert np.array_equal(np.array(altitudes), fx["altitudes"])
    np.testing.assert_allclose([ret, rel], [fx["ret"], fx["relative_return"]], rtol=1e-12)
    _check_curves(ct, entropies, f1s, fx)
    from test_oracle_golden import oracle_comatest_run
    _check_f1_counts(ct, oracle_comatest_run(fx)[3])
    # the GPU forward + argmax picks the recorded actions by itself (float32 conv on another device: allow a rare near-tie)
    assert np.mean(np.array(ct.greedy_actions) == fx["actions"]) >= 0.9
    ct2, out2 = run(False)
    if ct2.greedy_actions == list(fx["actions"]):
        np.testing.assert_allclose(out2[3], fx["entropies"], rtol=RTOL)


def test_batched_ig_policy_matches_oracle(name="small", over=None, seed=5, first_episode=40, n_envs=6):
    """VecEnv.ig_actions (K9 + K10 for all envs at once) against the oracle's literal restatement.
    (The arguments: tools/stress_parity.py sweeps random configurations through this same check.)"""
    from ippmarl.vec_env import VecEnv, POLICY_EXPLICIT
    params = make_params(name, **(over or {}))
    d = O.Derived(params)
    d.exact = True
    env = VecEnv(params, n_envs, philox_seed=seed)
    eps = np.arange(first_episode, first_episode + n_envs)
    env.reset(eps)
    for t in range(4):
        env.build_observations(t, features=False)
        # the beliefs the planner sees, as float64 probabilities of the stored log-odds (a float32 probability of a saturated cell,
        # 0.9999, keeps three digits of 1 - p, and a candidate over nothing but saturated cells has a gain made of exactly that)
        local = 1.0 / (1.0 + np.exp(-env.rows_view(env.local).cpu().numpy().astype(np.float64)))
        pos = env.pos.cpu().numpy()
        acts = env.ig_actions(communication=True)
        gains, chosen = env.ig_gains.cpu().numpy(), acts.cpu().numpy()
        for e in range(env.E):
            pls, gls, prior = [], [], []
            for i in range(d.n_agents):
                m = O.apply_collision_mask(d, pos[e, i], O.action_mask(d, pos[e, i]), prior)
                ap, g = O.ig_individual(d, pos[e, i], m, local[e, i])
                np.testing.assert_allclose(gains[e, i], g, rtol=RTOL, atol=1e-9)
                pls.append(ap), gls.append(g), prior.append(pos[e, i])
            util = O.ig_cell_utilities(pls, O.ig_relative(gls))
            for i, u in enumerate(util):
                want, got = int(np.argmax(u)), int(chosen[e][i])
                if want != got:
                    # Agents in contact hold identical fused maps, hence identical gains; two of them eyeing the same two cells
                    # crosswise get utilities p (1 - q) and (1 - p)(1 - (1 - q)) with q = p -- equal in exact arithmetic, told
                    # apart only by the rounding of 1 - (1 - p) (float64 in the reference, float32 here).  Only such ties may differ.
                    u = np.asarray(u, dtype=np.float64)
                    assert abs(u[want] - u[got]) <= 1e-6 * abs(u[want]), (t, e, i, u.tolist(), want, got)
        env.steps(t, policy=POLICY_EXPLICIT, actions=acts, features=False)


def test_state_and_metric_helpers_match_reference(golden):
    """utils.state.get_w_entropy_map / get_shannon_entropy and utils.utils.get_wrmse (the names IG_baseline.py:28-29 and
    coma_test.py:25-26 import) against the planes the reference itself returned (entropy_reward.npz)."""
    from ippmarl.agent.state_space import AgentStateSpace
    from ippmarl.utils.state import get_shannon_entropy, get_w_entropy_map
    from ippmarl.utils.utils import get_wrmse
    fx = golden("entropy_reward")
    params = make_params("small")
    ss = AgentStateSpace(params)
    for k in range(3):
        after, truth = fx[f"after{k}"], fx[f"truth{k}"].astype(np.float64)
        for mode in ("reward", "eval", "global"):
            arg = after.copy()
            wH, w, H, fp, grid = get_w_entropy_map(None, arg, truth, mode, ss)
            assert fp is None
            np.testing.assert_allclose(wH, fx[f"{mode}{k}_wH"], rtol=RTOL, atol=2e-6, err_msg=f"{mode} wH")
            np.testing.assert_array_equal(w, fx[f"{mode}{k}_w"])
            np.testing.assert_allclose(H, fx[f"{mode}{k}_H"], rtol=RTOL, atol=2e-6)
            np.testing.assert_allclose(grid, fx[f"{mode}{k}_p"], rtol=RTOL, atol=2e-7)
            assert w.dtype == fx[f"{mode}{k}_w"].dtype and wH.dtype == fx[f"{mode}{k}_wH"].dtype
            assert np.array_equal(arg, after)                       # the caller's map is copied / resized, never clipped
        p = after.copy()
        H = get_shannon_entropy(p, ss)
        assert p.min() >= np.float32(0.0001) and p.max() <= np.float32(0.9999)   # clipped in place (utils/state.py:118-121)
        np.testing.assert_allclose(H, fx[f"reward{k}_H"], rtol=RTOL, atol=2e-6)
    # actor mode: the footprint image goes through the same resize + weights
    rng = np.random.RandomState(4)
    m, fpimg = rng.random_sample((128, 128)).astype(np.float32), rng.random_sample((60, 60))
    wH, w, H, wfp, grid = get_w_entropy_map(fpimg.copy(), m, np.zeros((128, 128)), "actor", ss)
    q = O.area_resize(m.astype(np.float64), (11, 11))
    f = O.area_resize(fpimg, (11, 11))
    np.testing.assert_allclose(grid, np.clip(q, 1e-4, 0.9999), rtol=RTOL)
    np.testing.assert_allclose(wfp, O.class_weights(f) * O.shannon_entropy(f.copy()), rtol=RTOL, atol=2e-6)
    # F1 of the thresholded map (utils/utils.py:43-76) on a 128 x 128 map with values on both sides of, and exactly at, 0.5
    truth = (rng.random_sample((128, 128)) > 0.6).astype(np.float64)
    state = np.where(rng.random_sample((128, 128)) < 0.3, 0.5, m).astype(np.float32)
    state[:4, :4] = np.float32(0.5) + np.float32(2 ** -24)
    np.testing.assert_allclose(get_wrmse(state, truth, params), O.f1_target(state, truth), rtol=1e-12)
