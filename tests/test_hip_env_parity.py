"""GPU parity: the HIP path (through the C-ABI / ippmarl.VecEnv) against the oracle and the golden episodes.

Bit-exact: positions, footprint rects, action masks, actions, comm matrices, truth, measurement codes.
Floats (posteriors, rewards, features): within 1e-5 relative (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

import ipp_oracle as O
from configs import make_params
from conftest import REFERENCE_QUANTISATION_CELLS as RQ, assert_features_or_ties, assert_posteriors, local_subset, unpack_correctness
from test_oracle_golden import EPISODES

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _env(params, n_envs, **kw):
    from ippmarl.vec_env import VecEnv
    return VecEnv(params, n_envs, **kw)


def test_reset_matches_numpy_legacy_streams():
    params = make_params("small")
    d = O.Derived(params)
    env = _env(params, 64)
    eps = np.arange(1, 65) * 37 + 5
    env.reset(eps)
    pos = env.pos.cpu().numpy()
    for e, ep in enumerate(eps):
        for a in range(d.n_agents):
            assert list(pos[e, a]) == list(O.start_state(d, a, int(ep))), (ep, a)
        assert np.array_equal(env.truth_map[e].numpy(), O.make_truth(d, int(ep)).astype(np.uint8)), ep
    sp = env.split_pct.cpu().numpy()
    assert [tuple(v) for v in sp] == [O.truth_split_params(int(ep)) for ep in eps]


@pytest.mark.parametrize("tag", list(EPISODES))
def test_golden_episode_replay(golden, tag):
    """Replays the reference's recorded episode (its own flips, actions, comm draws) on the GPU."""
    from ippmarl.vec_env import POLICY_EXPLICIT
    fx = golden(tag)
    params = make_params(EPISODES[tag]["name"], **EPISODES[tag]["over"])
    d = O.Derived(params)
    n, T = d.n_agents, d.budget + 1
    corr = unpack_correctness(fx)
    comm = fx["comm_draws"].reshape(T, 1, n, n)
    env = _env(params, 1)
    dd = env.d

    def flips_for(stage, positions):
        rects = np.array([[dd.footprint(p)[1] for p in positions]])
        tiles = [[1 - corr[stage * n + i].reshape(rects[0, i, 3] - rects[0, i, 2], rects[0, i, 1] - rects[0, i, 0])
                  for i in range(n)]]
        return env.pack_flips(tiles, rects)

    env.reset([int(fx["episode"])], flips=flips_for(0, fx["positions"][0]))
    assert np.array_equal(env.truth_map[0].numpy(), fx["truth"])
    assert np.array_equal(env.pos[0].cpu().numpy(), fx["positions"][0])
    # (mapping.prior != 0.5: the reference re-quantises EVERY cell to a float32 probability at every fused message; its own
    #  recorded rewards and area averages carry that noise -- 1e-6 of S1 / S2, i.e. 2.2e-5 of the reward, and 1e-5 of an average)
    shifted = d.prior != 0.5
    fa, ra = (1e-5, 2.5e-5) if shifted else (2e-6, 1e-6)
    for t in range(T):
        obs = env.build_observations(t, comm_draws=torch.from_numpy(comm[t].copy()).to(env.device))
        np.testing.assert_allclose(obs[0].cpu().numpy(), fx["obs"][t], rtol=RTOL, atol=fa, err_msg=f"obs t={t}")
        acts = torch.from_numpy(fx["actions"][t][None].astype(np.int32))
        reward, done, state = env.steps(t, policy=POLICY_EXPLICIT, actions=acts, flips=flips_for(t + 1, fx["positions"][t + 1]))
        assert np.array_equal(env.mask[0].cpu().numpy(), fx["masks"][t].astype(np.uint8)), t
        assert np.array_equal(env.pos[0].cpu().numpy(), fx["positions"][t + 1]), t
        assert int(env.fault[0]) == 0
        np.testing.assert_allclose(float(reward[0, 0]), fx["rewards"][t, 0], rtol=RTOL, atol=ra, err_msg=f"reward t={t}")
        np.testing.assert_allclose(state[0].cpu().numpy(), fx["state"][t], rtol=RTOL, atol=fa, err_msg=f"state t={t}")
        assert done == bool(fx["done"][t, 0])
        if t == 0 and "global_t0" in fx:
            assert_posteriors(env.posterior_global()[0].cpu().numpy(), fx["global_t0"], strict=False, msg="global t=0", allow=RQ.get((tag, "global_t0"), []))
        if t == 7 and "global_t7" in fx:
            assert_posteriors(env.posterior_global()[0].cpu().numpy(), fx["global_t7"], strict=False, msg="global t=7", allow=RQ.get((tag, "global_t7"), []))
    assert_posteriors(local_subset(fx, env.posterior_local()[0].cpu().numpy()), fx["final_local"], strict=False, msg="final local", allow=RQ.get((tag, "final_local"), []))
    assert_posteriors(env.posterior_global()[0].cpu().numpy(), fx["final_global"], strict=False, msg="final global", allow=RQ.get((tag, "final_global"), []))


def _oracle_philox_episode(params, episode, seed, truth=None):
    """(log, final local maps, final global map) of one oracle episode under the device's randomness (tests/oracle_pool.py)."""
    from oracle_pool import philox_episode
    return philox_episode(params, episode, seed, truth)


# 85 x 85 cells: not a multiple of 4 wide AND 85^2 = 25 (mod 32), so the last cell's truth bit sits in the last-but-one byte group
# of the packed plane -- the 2-byte truth loads of the row-straddling groups reach the plane's end (ippm_truth_bytes keeps a spare byte)
EDGE_85 = ("default", dict(sensor__pixel__number_x=12, sensor__pixel__number_y=12, sensor__field_of_view__angle_x=70.0,
                           sensor__field_of_view__angle_y=70.0), 3)


@pytest.mark.parametrize("name,over,n_envs", [
    ("small", dict(), 6),
    ("small", dict(experiment__uav__failure_rate=0.35, experiment__uav__fix_range=False, experiment__missions__n_agents=6), 4),
    ("c2", dict(), 3),
    ("default", dict(experiment__missions__n_agents=3), 2),  # 493 cells: grid_y % 4 != 0 -> scalar path
    ("c4", dict(), 3),                                        # BASELINE config 4 shape: 8 UAVs, 512 x 512 (9-op plans), three envs
    ("c5", dict(experiment__missions__n_agents=3), 1),        # config 5 shape: 27 actions, 1024 x 1024, per-episode comm range
    ("small", dict(experiment__missions__n_agents=12, experiment__uav__communication_range=100), 1),  # >10 ops: generic fusion path
    ("small", dict(experiment__missions__n_agents=16, experiment__uav__communication_range=100,
                   experiment__constraints__num_actions=27), 1),                                        # the largest team the ABI admits
    ("small", dict(experiment__missions__n_agents=2, experiment__constraints__num_actions=9,
                   experiment__constraints__min_altitude=15, experiment__constraints__max_altitude=15), 2),  # smallest team, planar moves
    ("small", dict(experiment__missions__n_agents=3, experiment__constraints__num_actions=4), 3),    # the 2-D four-action set
    ("small", dict(mapping__prior=0.3), 3),   # prior != 0.5: every fusion shifts the whole grid (the explicit slow path)
    ("c2", dict(mapping__prior=0.45, experiment__missions__n_agents=3), 1),
    # (prior > 0.5 pulls every cell below 0.499 within two steps; all class weights are then 0 and the reference's relative
    #  reward is 0 / 0 = nan from there on: not a case to pin anything on)
    ("c5", dict(experiment__missions__n_agents=16), 2),  # two envs of BASELINE config 5 at its largest: 16 UAVs, 1024 x 1024, 27 actions
    # ... and its other team sizes ("mixed team sizes 2-16": the team size is a per-run parameter in the reference,
    # coma_wrapper.py:25-26, missions/episode_generator.py:99-102), two envs each, per-episode comm range, 27 actions
    ("c5", dict(experiment__missions__n_agents=2), 2),
    ("c5", dict(experiment__missions__n_agents=4), 2),
    ("c5", dict(experiment__missions__n_agents=8), 2),
    # altitudes beyond the sensor model's table (sensor_models.py:13-22: noise 0 unless z is 5 / 10 / 15 m): a measurement from
    # 20 m sets its cells to exactly 0 or 1, i.e. +-inf in log-odds storage, until the next fusion clips them
    ("small", dict(experiment__constraints__min_altitude=15, experiment__constraints__max_altitude=20,
                   experiment__constraints__num_actions=27, experiment__uav__communication_range=10), 2),
    EDGE_85,
])
def test_production_randomness_matches_oracle(name, over, n_envs):
    """Philox mode (what bench/training use): device RNG streams, uniform random policy, every step vs the oracle."""
    check_philox_episodes(name, over, n_envs)


def check_philox_episodes(name, over, n_envs, seed=0x1234567ABC, first_episode=11, track_area=True, fused_step=False, terrain="split",
                          team_sizes=None):
    """The every-step comparison of a batch with the oracle under the production randomness; returns the number of class-weight
    threshold ties met (conftest).  (``seed`` / ``first_episode``: tools/stress_parity.py sweeps random configurations through it.)
    ``fused_step``: ``steps()`` alone, i.e. ONE plan launch (comm + plans + work list + K1) -> fusion -> K3, the exact launch
    sequence bench.py times; the locally fused maps cannot be looked at between fusion and sensing then, so the local maps are
    compared after the step's sensing instead.  ``terrain="random_field"``: the device synthesises the field (bench.py's input);
    the generated truth is handed to the oracle, whose own field uses NumPy's legacy normal stream instead of Philox.
    ``team_sizes``: env e flies team_sizes[e] of the configured n_agents UAVs and is compared with an oracle run whose n_agents is
    team_sizes[e] (the reference's team size is a per-run parameter)."""
    from oracle_pool import philox_episodes
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params(name, **over)
    env = _env(params, n_envs, philox_seed=seed, track_area=track_area, terrain=terrain, team_sizes=team_sizes)
    eps = [first_episode + 7 * k for k in range(n_envs)]
    env.reset(eps)
    truths = None if terrain == "split" else list(env.truth_map.numpy().astype(np.float64))
    teams = [env.d.n_agents] * n_envs if team_sizes is None else [int(v) for v in team_sizes]
    oracles = philox_episodes(params if team_sizes is None else
                              [make_params(name, **dict(over, experiment__missions__n_agents=n_e)) for n_e in teams], eps, seed, truths)
    T = env.d.budget + 1
    feats = track_area and not fused_step   # (the 493 x 493 default grid included: its feature bins are not whole cells wide)
    ties = 0
    for t in range(T):
        if fused_step:
            obs = local = None
        else:
            obs = env.build_observations(t, features=feats)
            local = env.posterior_local().cpu().numpy()
        reward, done, state = env.steps(t, policy=POLICY_UNIFORM, features=feats)
        comm = env.comm.cpu().numpy()   # (written by the plan launch from the pre-move positions in either form)
        glob = env.posterior_global().cpu().numpy()
        sensed = env.posterior_local().cpu().numpy() if fused_step else None
        for e, (log, _, _) in enumerate(oracles):
            rec = log[t]
            n = teams[e]                      # the agents flying in this env (all of them unless team_sizes is given)
            want_comm = np.zeros((env.d.n_agents, env.d.n_agents), dtype=np.uint8)   # (nobody hears, or is heard by, the others)
            for i, ks in enumerate(rec["received"]):
                want_comm[i, ks] = 1
            assert np.array_equal(comm[e], want_comm), (t, e)
            assert np.array_equal(env.mask[e, :n].cpu().numpy(), rec["masks"].astype(np.uint8)), (t, e)
            assert np.array_equal(env.action[e, :n].cpu().numpy(), rec["actions"]), (t, e)
            assert np.array_equal(env.pos[e, :n].cpu().numpy(), rec["next_positions"]), (t, e)
            assert np.array_equal(env.rect[e, :n].cpu().numpy(), rec["next_rects"]), (t, e)
            assert not env.rect[e, n:].any(), (t, e)          # the others publish empty footprints
            # every cell of every map within 1e-5 -- prior != 0.5 included: there every message shifts every cell of the grid, and the
            # chain of a fusion runs in float64 registers and is rounded once, at the store
            strict = True
            if fused_step:
                assert_posteriors(sensed[e, :n], np.array(rec["sensed_local"]), strict=strict, msg=f"local after sensing t={t} e={e}")
            else:
                assert_posteriors(local[e, :n], np.array(rec["fused_local"]), strict=strict, msg=f"fused local t={t} e={e}")
            assert_posteriors(glob[e], rec["global_map"], strict=strict, msg=f"global t={t} e={e}")
            # returns: 1e-5 in every regime.  (Altitudes outside the sensor model's table are noise-free: cells jump between exactly
            # 0 / 1 and the clip, the reward terms are of size 1 with both signs and S1 is what is left after they cancel -- every
            # fusion kernel sums a row's / slot's cells in float32 and the lanes in float64; float32 lane sums showed at 2e-4 there.)
            noise_free = any(z not in (5, 10, 15) for z in env.d.altitudes)
            rt = RTOL
            # (the rewards are affine in the sums, 22 S1/S2 - 0.5 and 10 S1/cells - 0.17 (utils/reward.py:37-40): the tolerance of
            #  the sums applies to the part in front of the offset, which matters when a reward is close to 0)
            got_r = reward[e].cpu().numpy()
            np.testing.assert_allclose(got_r[0], rec["relative_reward"], rtol=rt, atol=1e-6 + rt * 0.5)
            np.testing.assert_allclose(got_r[1], rec["absolute_reward"], rtol=rt, atol=1e-6 + rt * 0.17)
            # (S1 = sum of w(a) (H(b) - H(a)) is a difference of two sums of the size of S2: its absolute error is a fraction of S2.
            #  Default: 2e-8 of S2.  Noise-free measurements: the terms are +-1 and nearly cancel, what remains is the float32 entropy
            #  of the saturated cells themselves, the same 3e-8 on each of them -- measured 5.4e-8 of S2 on an S1 of 0.83 next to an S2
            #  of 770: 2e-7.  prior != 0.5: every cell of the float32 maps enters both sums at every fusion with the 5e-7 absolute
            #  rounding of its stored log-odds: 1e-6 of S2.)
            s_scale = 1e-6 if env.d.prior != 0.5 else (2e-7 if noise_free else 2e-8)
            np.testing.assert_allclose(env.sums[e, :2].cpu().numpy(), [rec["s1"], rec["s2"]], rtol=rt, atol=1e-6 + s_scale * abs(rec["s2"]))
            if feats:   # (prior != 0.5: the area sums take a small change of EVERY cell at every fusion: 6e-6 absolute there)
                fa = 2e-6 if env.d.prior == 0.5 and not noise_free else 6e-6   # (noise-free: float32 increments of size 1/2)
                got_obs, got_state = obs[e].cpu().numpy(), state[e].cpu().numpy()
                for i in range(n):   # a differing element must be a proven class-weight threshold tie (conftest)
                    dec = {3: rec["decide_local"][i], 4: rec["decide_fp"][i]}
                    ties += assert_features_or_ties(got_obs[i], rec["observations"][i], dec, RTOL, fa, f"obs t={t} e={e} i={i}")
                    dec[8] = rec["decide_global"]
                    ties += assert_features_or_ties(got_state[i], rec["states"][i], dec, RTOL, fa, f"state t={t} e={e} i={i}")
    final = env.posterior_local().cpu().numpy()
    for e, (_, final_local, _) in enumerate(oracles):
        assert_posteriors(final[e, :teams[e]], final_local, strict=True, msg=f"final local e={e}")
        assert np.allclose(final[e, teams[e]:], env.d.prior, rtol=1e-6, atol=0), e   # the maps of agents that do not fly stay at the prior
    assert env.counters()["work_list_rejects"] == 0
    return ties


_MIXED_SMALL = ("small", dict(experiment__missions__n_agents=6), [1, 2, 3, 6, 4, 5])
_MIXED_SMALL27 = ("small", dict(experiment__missions__n_agents=5, experiment__uav__failure_rate=0.3, experiment__uav__fix_range=False,
                                experiment__constraints__num_actions=27), [5, 2, 4])
# BASELINE config 5 as it is worded: "mixed team sizes 2-16 UAVs, 3D altitude action space, 1024 x 1024 grid with comm-range
# masking" -- four envs of one batch flying 2, 4, 8 and 16 UAVs, each against an oracle run of that team size (once, in the
# env-only form: the tracked form is covered at 128 x 128)
_MIXED_C5 = ("c5", dict(experiment__missions__n_agents=16), [2, 4, 8, 16])


@pytest.mark.parametrize("name,over,teams,fused_step", [_MIXED_SMALL + (False,), _MIXED_SMALL + (True,), _MIXED_SMALL27 + (False,),
                                                        _MIXED_SMALL27 + (True,), _MIXED_C5 + (True,)])
def test_mixed_team_sizes_in_one_batch_match_oracle(name, over, teams, fused_step):
    """VecEnv(team_sizes=...): env e flies teams[e] of the configured UAVs and evolves, step by step, exactly like a run of the
    reference (oracle) whose n_agents is teams[e] -- comm matrices, masks, actions, positions, footprints bit for bit, maps, rewards
    and network inputs (agent-id plane (i + 1) / teams[e]) at 1e-5; with the network inputs built (two plan launches per step,
    tracked kernels) and in the env-only form bench.py times."""
    check_philox_episodes(name, over, len(teams), team_sizes=teams, fused_step=fused_step, track_area=not fused_step)


@pytest.mark.parametrize("name,over,n_envs", [
    ("small", dict(), 5), ("c2", dict(), 3), ("c4", dict(), 1),
    ("small", dict(experiment__missions__n_agents=12, experiment__uav__communication_range=100), 1),
    ("small", dict(mapping__prior=0.3), 2),
    ("small", dict(sensor__pixel__number_x=14, sensor__pixel__number_y=14, experiment__constraints__num_actions=27), 2),
    # noise-free altitudes (20 m is outside the sensor model's table): returns at 1e-5 in this form (float64 lane sums)
    ("small", dict(experiment__constraints__min_altitude=15, experiment__constraints__max_altitude=20,
                   experiment__constraints__num_actions=27, experiment__uav__communication_range=10), 2),
    EDGE_85,
    # 34 x 34 cells: narrower than 44, the one-cell-per-lane instantiations (k_sense_tiles<1>, the row walker's scalar form)
    ("default", dict(sensor__pixel__number_x=4, sensor__pixel__number_y=4, experiment__missions__n_agents=3), 4),
])
@pytest.mark.parametrize("fused_step", [False, True])
def test_untracked_env_step_matches_oracle(name, over, n_envs, fused_step):
    """The env-only step as bench.py runs it -- VecEnv(track_area=False): K3 in its tile form, the fusion in one-trip tile
    items from the plan kernel's work list -- through the same every-step comparison with the oracle (maps, masks, actions,
    rewards); ``fused_step`` = ``steps()`` alone, the single-plan-launch sequence of bench.py's timed loop."""
    check_philox_episodes(name, over, n_envs, seed=0x51C0FFEE11, first_episode=23, track_area=False, fused_step=fused_step)


@pytest.mark.parametrize("name,n_envs", [("c2", 3), ("small", 4)])
def test_benched_combination_matches_oracle(name, n_envs):
    """Exactly what bench.py's timed loop runs (BASELINE config 2): device-synthesised random-field terrain, no area sums
    (K3's tile form + the one-trip tile fusion), ``steps()`` alone = one plan launch per step -- every step against the oracle
    flying over the same generated field (mapping/ground_truths.py:25-40 for the field, coma_wrapper.py:73-183 for the step)."""
    check_philox_episodes(name, {}, n_envs, seed=3, first_episode=1, track_area=False, fused_step=True, terrain="random_field")


def test_class_weight_threshold_ties_are_proven_ties():
    """17 pixels per footprint: the footprint plane's area average lands on 0.501 exactly (0.5 + 0.125 * 20 / 2500) several
    times per episode (tools/find_tie_case.py: 4 such bins in episode 11 under this seed, 6 in episode 18).  The device may
    fall on either side there; the comparison admits a whole-class-weight difference only where the oracle's deciding average
    is within 32 ulp of the threshold, and everywhere else holds the usual 1e-5."""
    over = dict(sensor__pixel__number_x=17, sensor__pixel__number_y=17)
    params = make_params("small", **over)
    # the ties are there (so the rule is exercised whether or not the device happens to flip one of them)
    log, _, _ = _oracle_philox_episode(params, 11, 100)
    on_threshold = sum(int((np.minimum(np.abs(v - 0.499), np.abs(v - 0.501)) <= 32 * 2.0 ** -53).sum())
                       for rec in log for v in rec["decide_fp"])
    assert on_threshold >= 4
    ties = check_philox_episodes("small", over, 2, seed=100, first_episode=11)
    assert 0 <= ties <= 2 * on_threshold + 16   # (env 1 = episode 18 has its own)


@pytest.mark.parametrize("k", range(12))
def test_random_configurations_match_oracle(k):
    """A fixed dozen of the random configurations tools/stress_parity.py sweeps by the hundred (team size, action set, comm range,
    link failures, altitude lattice, grid size, prior, batch size all drawn at random) through the check above (17 pixels per
    footprint included: class-weight threshold ties are recognised as such by the comparison itself, see conftest)."""
    import random
    from random_configs import random_case
    name, over, n_envs, seed, ep0, _, _ = random_case(random.Random(7000 + k), pixels=(12, 13, 14, 16, 17, 18, 19))   # (11: footprint image < 11 cells)
    check_philox_episodes(name, over, min(n_envs, 3), seed=seed, first_episode=ep0, track_area=k % 2 == 0, fused_step=k % 4 == 3)


def _field_checks(got, want, tag):
    assert (want != got).mean() < 2e-3, (tag, (want != got).mean())   # float32 transform near the threshold
    assert 0.02 < got.mean() < 0.98, tag
    assert (got[1:, :] == got[:-1, :]).mean() > 0.95, tag              # k^-5 spectrum: large blobs


def test_random_field_terrain_fft_path():
    """Grids that are not powers of two (default 493 x 493, odd: one zero amplitude row/column): device noise + rocFFT +
    threshold kernel against a float64 host evaluation of the same noise."""
    from ippmarl import _ffi
    from ippmarl.terrain import amplitude_table
    params = make_params("default")
    E = 4
    env = _env(params, E, terrain="random_field")
    eps = np.array([3, 1000003, 17, 4])
    env.reset(eps)
    assert not env_terrain(env).native
    got = env.truth_map.numpy()
    gx, gy = env.d.grid_x, env.d.grid_y
    noise = torch.empty(E, gx, gy, dtype=torch.float32, device=env.device)
    env.ctx.call("ippm_terrain_noise", _ffi.ptr(env.episode), _ffi.ptr(noise), E, env.stream)
    z = noise.cpu().numpy().astype(np.float64)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    # Box-Muller over Philox(cell group; episode; terrain stream word), first group of env 1
    w = _ffi.host_philox(0, int(eps[1]) & 0xFFFFFFFF, 3 << 24, int(eps[1]) >> 32, 3, 0)
    u1, u2 = ((w[0] >> 8) + 1.0) / 2 ** 24, (w[1] >> 8) / 2 ** 24
    r = np.sqrt(-2.0 * np.log(u1))
    np.testing.assert_allclose(z[1].reshape(-1)[:2], [r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)], rtol=2e-5, atol=2e-6)
    amp = amplitude_table(gx, gy, params["sensor"]["simulation"]["cluster_radius"])
    for e in range(E):
        f = np.fft.ifft2(np.fft.fft2(z[e]) * amp).real
        f = (f - f.min()) / (f.max() - f.min())
        _field_checks(got[e], (f >= 0.5).astype(np.uint8), e)


def env_terrain(env):
    return env._field


@pytest.mark.parametrize("name", ["small", "c2", "c5"])   # 128, 256, 1024 cells a side
def test_random_field_terrain_native_path(name):
    """Power-of-two grids: spectrum drawn in the library + two-pass LDS inverse transform + threshold."""
    from ippmarl import _ffi
    from ippmarl.terrain import amplitude_table
    params = make_params(name, experiment__missions__n_agents=2)
    E = 3
    env = _env(params, E, terrain="random_field")
    eps = np.array([3, 1000003, 17])
    env.reset(eps)
    assert env_terrain(env).native
    got = env.truth_map.numpy()
    gx, gy = env.d.grid_x, env.d.grid_y
    hy = gy // 2 + 1
    amp = amplitude_table(gx, gy, params["sensor"]["simulation"]["cluster_radius"])[:, :hy]
    amp_dev = torch.from_numpy(np.ascontiguousarray(amp).astype(np.float32)).to(env.device)
    spec = torch.empty(E, gx, hy, 2, dtype=torch.float32, device=env.device)
    env.ctx.call("ippm_terrain_spectrum", _ffi.ptr(env.episode), _ffi.ptr(amp_dev), _ffi.ptr(spec), E, env.stream)
    S = spec.cpu().numpy().astype(np.float64)
    S = S[..., 0] + 1j * S[..., 1]
    # generic bins against the host Philox + Box-Muller; stream word = (stage 1, domain 3).  One call serves two bins: kx and
    # kx + gx/2 share call (kx mod gx/2) * hy + ky, the lower takes words (0, 1), the upper words (2, 3)
    for kx, ky in ((5, 9), (gx // 2 + 7, 3), (gx - 1, gy // 2 - 1)):
        w = _ffi.host_philox((kx % (gx // 2)) * hy + ky, int(eps[1]) & 0xFFFFFFFF, (1 << 8) | (3 << 24), int(eps[1]) >> 32, 3, 0)
        w1, w2 = (w[0], w[1]) if kx < gx // 2 else (w[2], w[3])
        u1, u2 = ((w1 >> 8) + 1.0) / 2 ** 24, (w2 >> 8) / 2 ** 24
        r = np.sqrt(-2.0 * np.log(u1)) * amp[kx, ky]
        # (the device uses the hardware log2 / sin / cos: a few float32 ulp of the radius off the libm values)
        np.testing.assert_allclose([S[1, kx, ky].real, S[1, kx, ky].imag], [r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)],
                                   rtol=1e-4, atol=3e-6 * r)
    # Hermitian structure of the self-mirrored columns; white-noise statistics of the generic bins
    for col in (0, gy // 2):
        assert np.array_equal(S[:, 1:gx // 2, col], np.conj(S[:, :gx // 2:-1, col]))
        assert np.all(S[:, [0, gx // 2], col].imag == 0)
    white = (S[:, 1:, 1:gy // 2] / amp[None, 1:, 1:gy // 2]).reshape(-1)
    assert abs(white.real.std() - 1) < 0.02 and abs(white.imag.std() - 1) < 0.02 and abs(white.mean()) < 0.02
    # inverse transform of an explicit spectrum against NumPy
    work = torch.empty(E, hy, gx, 2, dtype=torch.float32, device=env.device)
    field = torch.empty(E, gx, gy, dtype=torch.float32, device=env.device)
    keys = torch.empty(E, 2, dtype=torch.int32, device=env.device)
    env.ctx.call("ippm_terrain_field", None, None, _ffi.ptr(spec), _ffi.ptr(work), _ffi.ptr(field), _ffi.ptr(keys), E, env.stream)
    f_dev = field.cpu().numpy().astype(np.float64)
    for e in range(E):
        f = np.fft.irfft2(S[e], s=(gx, gy)) * (gx * gy)
        assert np.abs(f_dev[e] - f).max() < 2e-5 * np.abs(f).max(), (e, np.abs(f_dev[e] - f).max(), np.abs(f).max())
        fn = (f - f.min()) / (f.max() - f.min())
        _field_checks(got[e], (fn >= 0.5).astype(np.uint8), e)
    # the fused path (spectrum never stored) is the same arithmetic
    for rk in (None, keys):   # min/max reduced by the pack kernel, or taken from pass Y's atomics
        packed = torch.zeros_like(env.truth)
        env.ctx.call("ippm_terrain_pack", _ffi.ptr(field), _ffi.ptr(rk), _ffi.ptr(packed), E, env.stream)
        assert torch.equal(packed, env.truth)
    # an episode's terrain does not depend on the batch it is generated in
    solo = _env(params, 1, terrain="random_field")
    solo.reset([int(eps[1])])
    assert np.array_equal(solo.truth_map.numpy()[0], got[1])
    assert not np.array_equal(got[0], got[2])


@pytest.mark.parametrize("name,n_envs", [("c2", 3000), ("c4", 700), ("small", 501), ("c2", 7)])
def test_one_launch_terrain_equals_the_two_pass_form(name, n_envs):
    """ippm_terrain_truth's second transform pass as ONE launch (IPPM_TERRAIN_ONE_LAUNCH=1: the workgroups of an env exchange the
    field's (min, max) inside the launch -- arrival counter, workgroups numbered by a start-order ticket, every cross-workgroup word
    moved by device-scope read-modify-writes -- and threshold the rows they hold in registers) against the default two launches
    (min / max, then the same transforms again for the bits): the same truth bit for bit, at batches far larger than the device holds
    workgroups at once (3000 x 8 workgroups of 256 threads at 256^2; 32 workgroups per env at 512^2), twice in a row on the same
    scratch (the counters are re-armed by pass X), every workgroup arrived and no wait gave up (fault words 0).
    (1024^2 fields keep the two launches: with 128 workgroups per env spinning on one counter a wait ran into its bound once in a few runs.
    The form is not the default: measured, it gains nothing in the step -- DESIGN.md section 8; what it found is kept: the passes'
    complex products have a fixed contraction now, because the two launches of the default form used to disagree in a row's last
    bit once in a few hundred fields.)"""
    params = make_params(name, experiment__missions__n_agents=2)
    eps = np.arange(1, n_envs + 1) * 7919
    two = _env(params, n_envs, track_area=False, terrain="random_field")
    saved = os.environ.get("IPPM_TERRAIN_ONE_LAUNCH")
    os.environ["IPPM_TERRAIN_ONE_LAUNCH"] = "1"        # (read at ippm_ctx_create)
    try:
        one = _env(params, n_envs, track_area=False, terrain="random_field")
    finally:
        if saved is None:
            os.environ.pop("IPPM_TERRAIN_ONE_LAUNCH", None)
        else:
            os.environ["IPPM_TERRAIN_ONE_LAUNCH"] = saved
    for rep in range(2):
        ids = eps + rep
        one.reset(ids)
        two.reset(ids)
        assert torch.equal(one.truth, two.truth), (name, n_envs, rep)
        one.check_faults()
        keys = env_terrain(one)._keys.view(-1)[: 4 * n_envs].view(n_envs, 4)
        assert int(keys[:, 2].min()) == int(keys[:, 2].max()) == one.d.grid_x // {128: 32, 256: 32, 512: 16}[one.d.grid_x]   # every workgroup arrived
    assert int(one.truth.max()) > 0 and int(one.truth.min()) < 255      # (fields, not constants)


def test_terrain_prefetch_is_the_inline_terrain():
    """VecEnv.prefetch_terrain: the field of the next episodes synthesised on a side stream beside the current episodes' steps is
    bit for bit the field reset() synthesises in line; a reset with other ids than the prefetched ones ignores the prefetch."""
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params("c2")
    E = 48
    a = _env(params, E, terrain="random_field", track_area=False)
    b = _env(params, E, terrain="random_field", track_area=False)
    waves = [np.arange(1, E + 1) + 1000 * w for w in range(4)]
    for w, ids in enumerate(waves):
        a.reset(ids)
        b.reset(ids)
        if w + 1 < len(waves):
            # wave 1 is prefetched correctly, wave 2 with the WRONG ids (reset must fall back), wave 3 correctly again
            b.prefetch_terrain(waves[w + 1] if w != 1 else waves[w + 1] + 7)
        for t in range(a.d.budget + 1):
            ra, _, _ = a.steps(t, policy=POLICY_UNIFORM, features=False)
            rb, _, _ = b.steps(t, policy=POLICY_UNIFORM, features=False)
        torch.cuda.synchronize()
        assert torch.equal(a.truth, b.truth), w
        assert torch.equal(a.glob, b.glob) and torch.equal(a.local, b.local) and torch.equal(a.pos, b.pos), w
        assert not torch.equal(a.truth[0], a.truth[1])


def test_episode_on_random_field_terrain_matches_oracle():
    """Every step against the oracle flying over the same generated field (truth handed to both sides)."""
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params("small")
    seed = 77
    env = _env(params, 3, philox_seed=seed, terrain="random_field")
    eps = [5, 6, 7]
    env.reset(eps)
    truth = env.truth_map.numpy().astype(np.float64)
    oracles = [_oracle_philox_episode(params, ep, seed, truth=truth[e]) for e, ep in enumerate(eps)]
    for t in range(env.d.budget + 1):
        env.build_observations(t, features=False)
        reward, done, _ = env.steps(t, policy=POLICY_UNIFORM, features=False)
        glob = env.posterior_global().cpu().numpy()
        for e, (log, _, _) in enumerate(oracles):
            rec = log[t]
            assert np.array_equal(env.pos[e].cpu().numpy(), rec["next_positions"]), (t, e)
            assert_posteriors(glob[e], rec["global_map"], strict=True, msg=f"global t={t} e={e}")
            np.testing.assert_allclose(reward[e].cpu().numpy(), [rec["relative_reward"], rec["absolute_reward"]], rtol=RTOL, atol=1e-6)


def test_saturation_and_deferred_clamp():
    """Low-altitude hovering saturates cells past the clip bound: exercises the deferred full-grid clamp."""
    from ippmarl.vec_env import POLICY_EXPLICIT
    params = make_params("small", experiment__missions__n_agents=3, experiment__uav__communication_range=100)
    d = O.Derived(params)
    seed = 99
    # agents descend to 5 m and then shuttle between two cells so the same cells are observed many times
    script = [5, 5] + [4, 1] * 7
    starts = [[10, 10, 15], [15, 10, 15], [40, 40, 15]]

    def correctness(i, s, shape):
        _, fc = O.project_field_of_view(d, ep.agents[i]["position"])
        return O.philox_correctness(seed, 5, i, s, fc, d.gy, O.noise_of_altitude(ep.agents[i]["position"][2]))

    ep = O.OracleEpisode(params, 5, correctness, lambda i, t, m, o: script[t] if i < 2 else (3 if t % 2 == 0 else 2),
                         build_features=True, start_positions=starts, exact=True)
    log = ep.run()
    env = _env(params, 1, philox_seed=seed)
    env.reset([5], start_positions=torch.tensor([starts], dtype=torch.int32))
    exceeded = False
    for t in range(d.budget + 1):
        obs = env.build_observations(t)
        assert_posteriors(env.posterior_local()[0].cpu().numpy(), np.array(log[t]["fused_local"]), strict=True, msg=f"fused local t={t}")
        np.testing.assert_allclose(obs[0].cpu().numpy(), np.array(log[t]["observations"]), rtol=RTOL, atol=2e-6)
        acts = torch.from_numpy(np.asarray([log[t]["actions"]], dtype=np.int32))
        reward, _, state = env.steps(t, policy=POLICY_EXPLICIT, actions=acts)
        assert_posteriors(env.posterior_global()[0].cpu().numpy(), log[t]["global_map"], strict=True, msg=f"global t={t}")
        np.testing.assert_allclose(float(reward[0, 0]), log[t]["relative_reward"], rtol=RTOL, atol=1e-6)
        exceeded |= bool((env.local[0].abs() > env.d.logit_clip).any())
    assert exceeded, "scenario no longer saturates: the deferred-clamp path is not exercised"
    assert_posteriors(env.posterior_local()[0].cpu().numpy(), np.array([a["local_map"] for a in ep.agents]), strict=True, msg="final local")


@pytest.mark.parametrize("name,over,E,bench_form", [
    ("c2", dict(), 1024, False),   # BASELINE config 2 at full size, two plan launches per step (the training sequence), area sums tracked
    ("c2", dict(), 1024, True),    # ... and exactly as bench.py times it: track_area=False, steps() alone (tile-item fusion)
    ("c4", dict(), 1024, True),    # config 4's per-GPU shape: 1024 envs x 8 UAVs x 512^2 (9.4 GB of maps, 9-op plans)
    ("c5", dict(experiment__missions__n_agents=16), 64, True),  # a config 5 batch: 64 envs x 16 UAVs x 1024^2, 27 actions, per-episode comm range
])
def test_full_size_properties(name, over, E, bench_form):
    """Full BASELINE sizes: size-independent checks (the oracle takes seconds per env-step at these shapes)."""
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params(name, **over)
    env = _env(params, E, track_area=not bench_form)
    eps = np.arange(1, E + 1)
    small = _env(params, 8, track_area=not bench_form)
    pick = np.array([1, 2, 3, E // 2 - 12, E // 2 - 11, (3 * E) // 4 + 9, E - 24, E])
    d = env.d

    def episode(e):
        returns = torch.zeros(e.E, device=e.device)
        for t in range(d.budget + 1):
            if not bench_form:
                e.build_observations(t, features=False)
            r, _, _ = e.steps(t, policy=POLICY_UNIFORM, features=False)
            returns += r[:, 0]
        return returns

    small.reset(pick)
    env.reset(eps)
    returns = episode(env)
    small_returns = episode(small)
    assert int(env.fault.abs().sum()) == 0
    assert env.counters()["work_list_rejects"] == 0 and small.counters()["work_list_rejects"] == 0
    # (1) sharding independence: an episode's trajectory does not depend on the batch it runs in -- maps, positions and
    # measurement codes bit for bit; the returns to the summation order of the float64 reward atomics
    # (through rows_view: the large batch may live in tile storage, the small one in rows -- map_layout="auto" goes by the batch's size)
    assert torch.equal(env.rows_view(env.local[pick - 1]), small.rows_view(small.local))
    assert torch.equal(env.rows_view(env.glob[pick - 1]), small.rows_view(small.glob))
    assert torch.equal(env.pos[pick - 1], small.pos)
    assert torch.equal(env.code[pick - 1], small.code)
    torch.testing.assert_close(returns[pick - 1], small_returns, rtol=1e-6, atol=1e-6)
    # (2) the incrementally maintained weighted entropy T equals a full-grid recomputation
    full = torch.zeros(E, dtype=torch.float64, device=env.device)
    env.ctx.call("ippm_weighted_entropy", env._p(env.glob), None, 1, env._p(full), E, env.stream)
    torch.testing.assert_close(env.sums[:, 2], full, rtol=1e-6, atol=1e-3 * (d.grid_x / 256.0) ** 2)
    # (3) beliefs stay finite log-odds; posteriors are probabilities
    assert bool(torch.isfinite(env.local).all()) and bool(torch.isfinite(env.glob).all())
    pg = env.posterior_global()
    assert float(pg.min()) > 0.0 and float(pg.max()) < 1.0
    assert bool(torch.isfinite(returns).all())
    del pg
    # (4) positions stay on the lattice and inside the world
    p = env.pos.cpu().numpy()
    assert (p[..., :2] % d.spacing == 0).all() and p[..., :2].min() >= 0
    assert p[..., 0].max() <= d.x_dim_m and p[..., 1].max() <= d.y_dim_m
    assert set(np.unique(p[..., 2])) <= set(int(z) for z in d.altitudes)
    # (5) determinism: same episodes again -> identical bits
    env.reset(eps)
    episode(env)
    assert torch.equal(env.rows_view(env.glob[pick - 1]), small.rows_view(small.glob))
    assert torch.equal(env.rows_view(env.local[pick - 1]), small.rows_view(small.local))


def test_td_lambda_and_advantage_kernels(golden):
    from ippmarl import _ffi
    from ippmarl.derived import DerivedConstants
    fx = golden("td_lambda")
    dev = torch.device("cuda:0")
    ctx = _ffi.Context(DerivedConstants(make_params("c2")))
    stream = torch.cuda.current_stream().cuda_stream
    r = torch.tensor(fx["rewards"], dtype=torch.float32, device=dev)
    dn = torch.tensor(fx["dones"].astype(np.uint8), device=dev)
    qs = torch.tensor(fx["qsel"], dtype=torch.float32, device=dev)
    td, dr = torch.empty_like(r), torch.empty_like(r)
    ctx.call("ippm_td_lambda", r.data_ptr(), dn.data_ptr(), qs.data_ptr(), td.data_ptr(), dr.data_ptr(), r.shape[0], r.shape[1], stream)
    np.testing.assert_allclose(td.cpu().numpy(), fx["td"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dr.cpu().numpy(), fx["dr"], rtol=1e-5, atol=2e-6)
    r1, d1, q1 = r[:1, :15].contiguous(), dn[:1, :15].contiguous(), qs[:1, :15].contiguous()
    td1, dr1 = torch.empty_like(r1), torch.empty_like(r1)
    ctx.call("ippm_td_lambda", r1.data_ptr(), d1.data_ptr(), q1.data_ptr(), td1.data_ptr(), dr1.data_ptr(), 1, 15, stream)
    np.testing.assert_allclose(td1.cpu().numpy()[0], fx["td_single"], rtol=1e-5, atol=2e-6)
    # advantage
    rng = np.random.RandomState(4)
    B, A = 1000, 6
    probs = rng.dirichlet(np.ones(A), size=B).astype(np.float32)
    q = rng.standard_normal((B, A)).astype(np.float32)
    mask = (rng.random_sample((B, A)) > 0.3).astype(np.uint8)
    act = rng.randint(0, A, size=B).astype(np.int32)
    mask[np.arange(B), act] = 1
    mask[:3] = 0  # degenerate rows: all masked (floors at 1e-5 apply)
    adv_ref, _, pn_ref = O.coma_advantage(probs, q, mask.astype(np.float32), act)
    tp, tq, tm, ta = (torch.tensor(x, device=dev) for x in (probs, q, mask, act))
    adv = torch.empty(B, dtype=torch.float32, device=dev)
    pn = torch.empty(B, A, dtype=torch.float32, device=dev)
    ctx.call("ippm_coma_advantage", tp.data_ptr(), tq.data_ptr(), tm.data_ptr(), ta.data_ptr(), adv.data_ptr(), pn.data_ptr(), B, stream)
    np.testing.assert_allclose(adv.cpu().numpy(), adv_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pn.cpu().numpy(), pn_ref, rtol=1e-5, atol=1e-9)


def test_graph_replay_matches_eager():
    """hipGraph replay of {comm + plans + K1} -> {K4 + K5} gives bit-identical state to the eager launch sequence."""
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params("small")
    a, b = _env(params, 16), _env(params, 16)
    eps = np.arange(21, 37)
    a.reset(eps)
    b.reset(eps)
    b.capture_step_graphs(POLICY_UNIFORM)
    for wave in range(2):
        for t in range(a.d.budget + 1):
            a.build_observations(t, features=False)
            ra, _, _ = a.steps(t, policy=POLICY_UNIFORM, features=False)
            rb, _ = b.step_graphed(t)
            assert torch.equal(a.pos, b.pos) and torch.equal(a.action, b.action) and torch.equal(a.mask, b.mask), (wave, t)
            assert torch.equal(ra, rb), (wave, t)
            # the step's work list (a replay must rebuild it, not append to it): the same items per env -- in the tile form their
            # order within an env's slice is whatever the builders' LDS atomics made it
            E = a.E
            assert torch.equal(a.work[:E], b.work[:E]), (wave, t)
            wa, wb = a.work.cpu().numpy(), b.work.cpu().numpy()
            d = a.d   # (ippm_tile_env_cap: items an env's slice holds)
            ops, G = d.n_agents + 1, (d.grid_y + 3) // 4
            slots = 4 if ops <= 4 else (2 if ops <= 10 else 1)
            base, cap = (E + 3) & ~3, ops * (-(-d.grid_x * G // (32 * slots)) + ops * (2 * ops - 1) * ((G + 63) // 64))
            assert base + E * cap * 4 <= len(wa)
            for e in (0, 7, E - 1):
                n_items = int(wa[e]) & 0x0FFFFFFF
                assert int(wa[e]) & 0x40000000 and 0 < n_items <= cap
                ia = wa[base + e * cap * 4: base + (e * cap + n_items) * 4].reshape(-1, 4)
                ib = wb[base + e * cap * 4: base + (e * cap + n_items) * 4].reshape(-1, 4)
                assert sorted(map(tuple, ia)) == sorted(map(tuple, ib)), (wave, t, e)
        assert torch.equal(a.local, b.local) and torch.equal(a.glob, b.glob)
        a.reset(eps + 100)
        b.reset(eps + 100)


def test_work_list_form_follows_the_context():
    """A C caller that plans WITHOUT IPPM_STEP_TILES and fuses with (or without) area sums gets the same maps as one that passes
    the flag: the form of the work list is the context's (include/ippmarl.h), so plan and fusion cannot disagree.  (ADVICE r04:
    until round 5 such a caller's list was skipped by the tile fusion -- no fusion, no area update, rc 0.)"""
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params("small")
    for track in (True, False):
        a, b = _env(params, 8, track_area=track), _env(params, 8, track_area=track)
        assert a._tile_form
        b._tile_form = False          # VecEnv then never sets the flag; the library must build the tile form all the same
        eps = np.arange(5, 13)
        a.reset(eps)
        b.reset(eps)
        for t in range(a.d.budget + 1):
            a.build_observations(t, features=False)
            b.build_observations(t, features=False)
            a.steps(t, policy=POLICY_UNIFORM, features=False)
            b.steps(t, policy=POLICY_UNIFORM, features=False)
            assert torch.equal(a.local, b.local) and torch.equal(a.glob, b.glob) and torch.equal(a.pos, b.pos), (track, t)
            assert int(b.work[0]) & 0x40000000, "the list was not written in the tile form"
        assert float((a.glob != 0).float().mean()) > 0.2     # the global maps were fused at all
        if track:
            np.testing.assert_allclose(a.area.cpu().numpy(), b.area.cpu().numpy(), rtol=1e-12)
        assert a.counters()["work_list_rejects"] == 0 and b.counters()["work_list_rejects"] == 0


def test_split_batch_on_two_streams_equals_the_batch():
    """SplitVecEnv: the batch stepped as sub-batches on their own HIP streams (bench.py's default: the plan kernel and the resets
    of one half run beside the map kernels of the other) computes, bit for bit, what one VecEnv computes for the same episodes --
    uneven split, mixed team sizes and a reset in the middle included."""
    from ippmarl.vec_env import SplitVecEnv, POLICY_UNIFORM
    params = make_params("small", experiment__uav__fix_range=False, experiment__uav__failure_rate=0.2)
    teams = [4, 2, 3, 4, 1, 4, 2]
    one = _env(params, 7, track_area=False, terrain="random_field", team_sizes=teams)
    for parts in (2, 3):
        split = SplitVecEnv(params, 7, parts=parts, terrain="random_field", team_sizes=teams)
        assert split.sizes == ([4, 3] if parts == 2 else [3, 2, 2])
        for wave in range(2):
            eps = np.arange(5, 12) + 50 * wave
            one.reset(eps)
            split.reset(eps)
            for t in range(one.d.budget + 1):
                r1, _, _ = one.steps(t, policy=POLICY_UNIFORM, features=False)
                split.steps(t, policy=POLICY_UNIFORM)
                assert torch.equal(split.pos, one.pos) and torch.equal(split.action, one.action), (parts, wave, t)
                assert torch.equal(split.reward, r1), (parts, wave, t)
            assert torch.equal(split.local, one.local) and torch.equal(split.glob, one.glob), (parts, wave)
        got, want = split.counters(), one.counters(reset=True)
        assert got == want, (got, want)
        assert int(split.fault.abs().sum()) == 0


@pytest.mark.parametrize("name,over,n_envs,slabs", [
    ("c2", {}, 40, True), ("c2", {}, 40, False),
    ("small", dict(experiment__missions__n_agents=5, experiment__constraints__num_actions=27), 33, True),
    ("default", dict(experiment__missions__n_agents=2), 6, True),
    ("c4", dict(experiment__uav__fix_range=False), 9, True), ("c4", dict(experiment__uav__fix_range=False), 5, False)])
def test_reset_leaves_nothing_of_the_last_episode(name, over, n_envs, slabs, monkeypatch):
    """The env-only reset writes the prior only where the finished episode wrote (per 16-row dirty slab of every map, marked by the
    plan kernel -- ippm_set_dirty_slabs, IPPM_DIRTY_SLABS=1 -- or, the default, one bounding box per map) and senses the start footprints in the same
    launch.  After every reset of three waves of random-policy episodes: every cell of every local map outside its agent's start
    footprint is EXACTLY the prior (log-odds 0), every cell of the global maps is, the cells inside the footprints are the two
    measurement log-odds, and the slab records are re-armed to exactly the start footprints.  (A cell the fill missed would carry the
    last episode's belief into the next one; the oracle comparisons of the other tests would see it only where a footprint meets it.)"""
    from ippmarl.vec_env import POLICY_UNIFORM
    from ippmarl import _ffi
    monkeypatch.setenv("IPPM_DIRTY_SLABS", "1" if slabs else "0")
    params = make_params(name, **over)
    env = _env(params, n_envs, track_area=False, terrain="random_field" if name != "default" else "split")
    assert (env.slabs is not None) == slabs
    d = env.d
    T = d.budget + 1
    for wave in range(3):
        env.reset(np.arange(1, n_envs + 1) + 1000 * wave)
        rect = env.rect.cpu().numpy()                      # [E, N, (yu, yd, xl, xr)] of the start positions
        local = env.rows_view(env.local).cpu().numpy()
        assert not env.glob.ne(0).any(), wave
        inside = np.zeros(local.shape, dtype=bool)
        for e in range(n_envs):
            for i in range(d.n_agents):
                yu, yd, xl, xr = rect[e, i]
                inside[e, i, xl:xr, yu:yd] = True
        assert not (local[~inside] != 0).any(), (wave, int((local[~inside] != 0).sum()))
        vals = np.unique(local[inside])
        assert len(vals) <= 2 and np.all(vals != 0), vals       # (all UAVs start at 15 m: one pair of measurement log-odds)
        if slabs:
            ns = (d.grid_x + 15) // 16
            sl = env.slabs.view(n_envs, d.n_agents + 1, 2, ns).cpu().numpy()
            assert np.all(sl[:, d.n_agents, 0] == 0x7FFFFFFF) and np.all(sl[:, d.n_agents, 1] == 0)      # global maps: nothing yet
            for e in range(n_envs):
                for i in range(d.n_agents):
                    yu, yd, xl, xr = rect[e, i]
                    for k in range(ns):
                        meets = xl < 16 * k + 16 and xr > 16 * k and xr > xl and yd > yu
                        assert (sl[e, i, 0, k], sl[e, i, 1, k]) == ((yu, yd) if meets else (0x7FFFFFFF, 0)), (wave, e, i, k)
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        if slabs:     # what the episode marked covers what it wrote
            sl = env.slabs.view(n_envs, d.n_agents + 1, 2, -1).cpu().numpy()
            maps = np.concatenate([env.rows_view(env.local).cpu().numpy(), env.rows_view(env.glob).cpu().numpy()[:, None]], axis=1) != 0      # [E, N+1, gx, gy]
            cols = np.arange(d.grid_y)
            for k in range(sl.shape[-1]):
                written = maps[:, :, 16 * k:16 * k + 16].any(axis=2)                                           # [E, N+1, gy]
                marked = (cols[None, None] >= sl[:, :, 0, k, None]) & (cols[None, None] < sl[:, :, 1, k, None])
                assert not (written & ~marked).any(), (wave, k)
        env.check_faults()


@pytest.mark.parametrize("parts", [2, 3])
def test_staggered_sub_batches_fly_the_same_episodes(parts):
    """SplitVecEnv.start / advance (bench.py's default loop): every part in its own phase of the episode -- part k runs k * T / parts
    steps ahead, so that at most one part resets at any step -- flies, bit for bit, the episodes one VecEnv flies: after any number
    of advances, part k's envs hold wave w_k of their episodes at step t_k, exactly as a whole-batch VecEnv reset to wave w_k and
    stepped t_k times holds them (maps, positions, actions, rewards; uneven split, mixed team sizes, resets in the middle)."""
    from ippmarl.vec_env import SplitVecEnv, POLICY_UNIFORM
    params = make_params("small", experiment__uav__fix_range=False, experiment__uav__failure_rate=0.2)
    teams = [4, 2, 3, 4, 1, 4, 2]
    E = len(teams)
    ids = lambda wave: np.arange(5, 5 + E) + 50 * wave      # noqa: E731
    one = _env(params, E, track_area=False, terrain="random_field", team_sizes=teams)
    split = SplitVecEnv(params, E, parts=parts, terrain="random_field", team_sizes=teams)
    T = one.d.budget + 1
    split.start(ids, stagger=True)
    assert split._phase == [k * T // parts for k in range(parts)] and split.part_resets == parts
    done = 0
    for advances in (1, T // 2, T):                  # looked at after 1, T/2 + 1 and 3T/2 + 1 advances: every part passes a reset
        for _ in range(advances):
            split.advance()
        done += advances
        split.join()
        for k, (env, off, n) in enumerate(zip(split.parts, split.offsets, split.sizes)):
            ahead = k * T // parts + done
            assert (split._wave[k], split._phase[k]) == (ahead // T, ahead % T)
            one.reset(ids(split._wave[k]))
            r1 = None
            for t in range(split._phase[k]):
                r1, _, _ = one.steps(t, policy=POLICY_UNIFORM, features=False)
            sl = slice(off, off + n)
            assert torch.equal(env.episode, one.episode[sl]) and torch.equal(env.pos, one.pos[sl]), (k, done)
            assert torch.equal(env.rows_view(env.local), one.rows_view(one.local[sl])), (k, done)
            assert torch.equal(env.rows_view(env.glob), one.rows_view(one.glob[sl])), (k, done)
            assert torch.equal(env.rect, one.rect[sl]), (k, done)      # (the code plane keeps stale bytes outside the current footprints)
            if r1 is not None:
                assert torch.equal(env.action, one.action[sl]) and torch.equal(env.reward, r1[sl]), (k, done)
    assert split.part_resets == parts + sum(split._wave)
    assert int(split.fault.abs().sum()) == 0


@pytest.mark.parametrize("prior", [0.5, 0.3])
def test_tile_fusion_xcd_rotation_changes_nothing(prior):
    """The tile fusion deals an env's wavefronts out over the eight XCDs by rotating the env index with the wavefront index when the
    batch is even and at least 8 envs (fuse_tiles.hip); odd or smaller batches launch unrotated.  Which wavefront does which item
    must not show: a batch of 16 envs (rotated) against the same episodes as sub-batches of 6 + 5 + 5 (unrotated), and against
    IPPM_TILE_ROTATE=0, bit for bit -- maps, rewards, work counters.  prior 0.3: the row walker's work list (fuse.hip), whose
    wavefronts are rotated the same way (no knob there: the third env repeats the first)."""
    from ippmarl.vec_env import SplitVecEnv, POLICY_UNIFORM
    params = make_params("small", experiment__uav__fix_range=False, experiment__uav__failure_rate=0.1, experiment__missions__n_agents=5,
                         mapping__prior=prior)
    teams = [5, 2, 3, 5, 1, 4, 5, 5] * 2
    one = _env(params, 16, track_area=False, terrain="random_field", team_sizes=teams)
    split = SplitVecEnv(params, 16, parts=3, terrain="random_field", team_sizes=teams)
    assert split.sizes == [6, 5, 5]
    saved = os.environ.get("IPPM_TILE_ROTATE")
    os.environ["IPPM_TILE_ROTATE"] = "0"      # (read at ippm_ctx_create)
    try:
        plain = _env(params, 16, track_area=False, terrain="random_field", team_sizes=teams)
    finally:
        if saved is None:
            os.environ.pop("IPPM_TILE_ROTATE", None)
        else:
            os.environ["IPPM_TILE_ROTATE"] = saved
    eps = np.arange(21, 37)
    for env in (one, split, plain):
        env.reset(eps)
    for t in range(one.d.budget + 1):
        r1, _, _ = one.steps(t, policy=POLICY_UNIFORM, features=False)
        split.steps(t, policy=POLICY_UNIFORM)
        r3, _, _ = plain.steps(t, policy=POLICY_UNIFORM, features=False)
        assert torch.equal(split.pos, one.pos) and torch.equal(plain.pos, one.pos), t
        assert torch.equal(split.reward, r1) and torch.equal(r3, r1), t
    assert torch.equal(split.local, one.local) and torch.equal(split.glob, one.glob)
    assert torch.equal(plain.local, one.local) and torch.equal(plain.glob, one.glob)
    want = one.counters(reset=True)
    assert split.counters() == want and plain.counters(reset=True) == want
    assert want["fuse_local_cells"] > 0 and want["work_list_rejects"] == 0


def test_split_streams_run_side_by_side():
    """SplitVecEnv checks that its streams do not share a hardware queue (HIP serves a process's streams from four; two torch
    streams in eleven did on the box of tools/stream_queue_probe.py) and swaps the ones that do: afterwards a spin kernel on each of two
    parts' streams takes about the time of one, not of two."""
    from ippmarl.vec_env import SplitVecEnv
    if not hasattr(torch.cuda, "_sleep"):
        pytest.skip("no spin kernel in this torch build")
    keep = [torch.cuda.Stream() for _ in range(9)]     # a process with other streams alive, as bench.py's is
    split = SplitVecEnv(make_params("small"), 6, parts=3, terrain="random_field")
    assert len(split.stream_probe) == 2 and max(split.stream_probe) < 0.8, (split.stream_redraws, split.stream_probe)
    for j in range(3):
        for k in range(j + 1, 3):
            ratio, serial = SplitVecEnv._side_by_side(split.streams[j], split.streams[k], 1 << 20)
            assert serial < 4e-4 or ratio < 0.8, (j, k, ratio, serial)
    del keep


def test_fused_comm_and_plan_equals_separate_calls():
    """ippm_comm_fuse_local == ippm_comm_matrix + ippm_fuse_local (bitwise), incl. link failures and per-episode ranges."""
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params("small", experiment__uav__failure_rate=0.3, experiment__uav__fix_range=False, experiment__missions__n_agents=5)
    a, b = _env(params, 24), _env(params, 24)
    eps = np.arange(3, 27)
    a.reset(eps)
    b.reset(eps)
    for t in range(a.d.budget + 1):
        a.build_observations(t, features=False)          # fused entry point
        b.comm_matrix(t)
        b.fuse_local()
        # (workspace words 6, 7 and 14, 15 = the written-cells boxes, which only the batched step's plan kernel keeps)
        assert torch.equal(a.comm, b.comm) and torch.equal(a.local, b.local), t
        for lo, hi in ((0, 6), (8, 14), (16, None)):
            assert torch.equal(a.ws[:, :-1, lo:hi], b.ws[:, :-1, lo:hi]), (t, lo)
        a.steps(t, policy=POLICY_UNIFORM, features=False)
        # b: the stand-alone K5 entry point (plan + fusion + finalize), then K1 and K3 through the step's own kernels
        b.ctx.call("ippm_fuse_global_reward", b._p(b.glob), b._p(b.code), b._p(b.rect), b._p(b.pos), b._p(b.ws), b._p(b.sums),
                   b._p(b.reward), b.E, b.stream)
        rb = b.reward.clone()
        b.ctx.call("ippm_mask_act_move", b._p(b.episode), b._p(b.pos), None, None, POLICY_UNIFORM, t, b._p(b.mask), b._p(b.action),
                   b._p(b.fault), b.E, b.stream)
        b.sense(stage=t + 1)
        assert torch.equal(a.pos, b.pos) and torch.equal(a.glob, b.glob) and torch.equal(a.local, b.local), t
        assert torch.equal(a.reward, rb) and torch.equal(a.rect, b.rect), t


@pytest.mark.parametrize("name,over,n_envs", [("small", {}, 6), ("c2", {}, 3), ("default", {"experiment__missions__n_agents": 3}, 2),
                                              # 20 m: the reference's sensor is noise-free there -> infinite log-odds in the maps
                                              ("small", {"experiment__constraints__min_altitude": 15, "experiment__constraints__max_altitude": 20,
                                                         "experiment__constraints__num_actions": 27}, 3),
                                              ("small", {"experiment__constraints__num_actions": 27, "experiment__missions__n_agents": 9}, 2)])
def test_tracked_area_sums_equal_a_streaming_recomputation(name, over, n_envs):
    """The 11x11 area sums K3 / K4 / K5 maintain incrementally (K6's only view of the maps) against ippm_area_sums' full
    streaming pass over the same maps after every step, and the streaming pass against NumPy (exact area weights)."""
    from ippmarl import _ffi
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params(name, **over)
    env = _env(params, n_envs, philox_seed=11)
    d = env.d
    N, G = d.n_agents, d.grid_x * d.grid_y
    env.reset(np.arange(40, 40 + n_envs))
    fresh = torch.zeros_like(env.area)

    def check(tag):
        env.ctx.call("ippm_area_sums", env._p(env.local), _ffi.ptr(fresh), env.E * N, N, 0, env.stream)
        env.ctx.call("ippm_area_sums", env._p(env.glob), _ffi.ptr(fresh), env.E, 1, N, env.stream)
        got, want = env.area.cpu().numpy() / G, fresh.cpu().numpy() / G
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-7, err_msg=tag)   # area averages lie in [0, 1]

    check("reset")
    for t in range(d.budget + 1):
        env.build_observations(t, features=False)
        check(f"fusion of step {t}")
        env.steps(t, policy=POLICY_UNIFORM, features=False)
        check(f"sensing of step {t}")
    # the streaming pass itself against the exact area average of the exported probabilities
    W = O.area_weights(d.grid_x, 11), O.area_weights(d.grid_y, 11)
    p = env.posterior_local().cpu().numpy().astype(np.float64)
    want = np.einsum("ax,enxy,by->enab", W[0], p, W[1])
    np.testing.assert_allclose(fresh[:, :N].cpu().numpy().reshape(env.E, N, 11, 11) / G, want, rtol=0, atol=3e-7)


def test_c_abi_error_paths():
    """Error behaviour of the C-ABI: negative return codes with a message, never a crash."""
    import ctypes as C
    from ippmarl import _ffi
    from ippmarl.derived import DerivedConstants
    lib = _ffi.load_library()
    d = DerivedConstants(make_params("small"))
    # rejected configurations
    for mutate, needle in ((lambda c: setattr(c, "prior", 1.5), "prior"), (lambda c: setattr(c, "n_agents", 40), "n_agents"),
                           (lambda c: setattr(c, "n_actions", 5), "num_actions"), (lambda c: setattr(c, "tile_stride", 8), "tile_stride"),
                           (lambda c: setattr(c, "x_dim_m", 1 << 20), "x_dim"), (lambda c: setattr(c, "spacing", 20000), "spacing"),
                           (lambda c: setattr(c, "min_altitude", -5), "altitudes")):
        cfg = _ffi.make_config(d)
        mutate(cfg)
        h = C.c_void_p()
        rc = lib.ippm_ctx_create(C.byref(cfg), C.byref(h))
        assert rc < 0 and needle in lib.ippm_last_error().decode(), (rc, lib.ippm_last_error())
    ctx = _ffi.Context(d)
    stream = torch.cuda.current_stream().cuda_stream
    with pytest.raises(_ffi.IppmError, match="null argument"):
        ctx.call("ippm_footprint", None, None, None, 1, stream)
    with pytest.raises(_ffi.IppmError, match="Philox"):
        z = torch.zeros(16, device="cuda")
        ctx.call("ippm_sense_update", None, z.data_ptr(), z.data_ptr(), z.data_ptr(), None, z.data_ptr(), z.data_ptr(), None, 0, -1, 0, stream)
    with pytest.raises(_ffi.IppmError, match="policy"):
        ctx.call("ippm_mask_act_move", None, z.data_ptr(), None, None, 7, 0, z.data_ptr(), z.data_ptr(), None, 0, stream)
    with pytest.raises(_ffi.IppmError, match="agent_sel"):
        ctx.call("ippm_fuse_local", z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 99, 0, stream)
    # the feature kernels need the 11 x 11 lattice the networks are wired for
    p = make_params("small", environment__x_dim=40)
    env = _env(p, 1)
    env.reset([1])
    with pytest.raises(_ffi.IppmError, match="11x11"):
        env.build_observations(0)
    ctx.close()


def test_empty_mask_sets_fault_flag():
    """All actions blocked (where the reference's torch.multinomial raises): fault flag + a boundary-valid move."""
    from ippmarl.vec_env import POLICY_SAMPLE
    params = make_params("small")
    d = O.Derived(params)
    env = _env(params, 2)
    # env 0: agent 3 sits in the corner column (0,0); the three agents moving before it end up at (5,0), (0,5) and in its
    # own column -> +x, +y and both vertical moves get masked (action_space.py:328-344), nothing is left.  env 1: harmless.
    start = torch.tensor([[[10, 0, 15], [0, 10, 15], [0, 0, 15], [0, 0, 10]],
                          [[25, 25, 15], [30, 30, 15], [10, 40, 10], [40, 10, 5]]], dtype=torch.int32)
    env.reset([1, 2], start_positions=start)
    want = [[1, 2, 5, 0], [4, 4, 4, 4]]
    probs = torch.zeros(2, 4, 6)
    for e in range(2):
        for i in range(4):
            probs[e, i, want[e][i]] = 1.0
    env.build_observations(0, features=False)
    env.steps(0, policy=POLICY_SAMPLE, probs=probs.to(env.device), features=False)
    assert env.fault.cpu().tolist() == [1 << 3, 0]
    assert env.mask[0, 3].cpu().tolist() == [0, 0, 0, 0, 0, 0]
    p = env.pos.cpu().numpy()
    assert p[0, :3].tolist() == [[5, 0, 15], [0, 5, 15], [0, 0, 10]]
    assert p[0, 3].tolist() == [0, 0, 15]          # first boundary-valid action (up) keeps the state on the lattice
    assert p[1].tolist() == [[30, 25, 15], [35, 30, 15], [15, 40, 10], [45, 10, 5]]
    # the oracle raises in the same situation, like torch.multinomial in the reference
    m = O.action_mask(d, np.array([0, 0, 10]))
    m = O.apply_collision_mask(d, np.array([0, 0, 10]), m, [np.array([5, 0, 15]), np.array([0, 5, 15]), np.array([0, 0, 10])])
    assert m.sum() == 0
    with pytest.raises(ValueError):
        O.uniform_valid_action(123, m)


def test_sense_records_hold_footprint_and_sensor_constants():
    """K1's sense records (rect_next, what K3 starts from): the footprint K3 then publishes, and the measurement log-odds (minus
    logit(prior), float32 arithmetic) and flip threshold of the agent's NEW altitude, bit for bit from the host tables."""
    from ippmarl import _ffi
    from ippmarl.vec_env import POLICY_UNIFORM
    for name, over in (("c2", {}), ("small", {"mapping__prior": 0.45})):
        params = make_params(name, **over)
        env = _env(params, 16, track_area=(name == "small"))
        d = env.d
        env.reset(np.arange(5, 21))
        for t in range(4):
            if env.track_area:
                env.build_observations(t)
            env.steps(t, policy=POLICY_UNIFORM, features=False)
            torch.cuda.synchronize()
            rec = env.rect_next.cpu().numpy()
            assert rec.shape[-1] == _ffi.SENSE_REC_WORDS
            np.testing.assert_array_equal(rec[..., :4], env.rect.cpu().numpy())
            k = (env.pos.cpu().numpy()[..., 2] - d.min_altitude) // d.spacing
            lp = np.float32(d.logit_prior)
            want_lm = (d.logit_meas[k] - lp).astype(np.float32)                     # [E, N, 2]
            np.testing.assert_array_equal(rec[..., 4:6].view(np.float32), want_lm)
            np.testing.assert_array_equal(rec[..., 6].astype(np.int64) & 0xFFFFFFFF, d.flip_threshold[k].astype(np.int64))
            assert not rec[..., 7].any()


def test_placement_search_changes_addresses_not_results():
    """VecEnv.tune_placement: the hot planes move to the allocation the map kernels ran fastest on; an episode afterwards is bit
    for bit the episode of an env that never searched."""
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params("c2")
    E = 256                                                       # the smallest batch the search bothers with
    a, b = _env(params, E, track_area=False), _env(params, E, track_area=False)
    before = {n: getattr(b, n).data_ptr() for n in ("local", "glob", "code", "truth")}
    res = b.tune_placement(3)
    assert res is not None and 1 <= res["draws"] <= 3 and len(res["map_kernels_us_per_step"]) == res["draws"]
    assert res["map_kernels_us_per_step"][res["kept"]] == min(res["map_kernels_us_per_step"])
    if res["kept"] != 0:
        assert all(getattr(b, n).data_ptr() != before[n] for n in before)
    assert _env(params, 8, track_area=False).tune_placement(3) is None   # small batches: nothing to search for
    ids = np.arange(11, 11 + E)
    for env in (a, b):
        env.reset(ids)
        for t in range(env.d.budget + 1):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
    torch.cuda.synchronize()
    for n in ("local", "glob", "pos", "truth", "reward", "action"):   # (code tiles keep bytes of earlier footprints outside the current one)
        assert torch.equal(getattr(a, n), getattr(b, n)), n


def test_kernel_timing_reports_dispatch_durations():
    """ippm_kernel_timing / ippm_read_kernel_times (what bench.py's roofline leg reads): every launch of the timed classes made
    while timing is on is counted once, under the name the compiler gives the kernel, with a plausible begin-to-end duration;
    launches made while it is off are not."""
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params("c2")
    env = _env(params, 64, track_area=False)
    env.reset(np.arange(1, 65))
    env.steps(0, policy=POLICY_UNIFORM, features=False)          # not timed
    env.profile = True
    for t in range(1, 6):
        env.steps(t, policy=POLICY_UNIFORM, features=False)
    env.profile = False
    env.steps(6, policy=POLICY_UNIFORM, features=False)          # not timed
    times = env.event_times_us()
    assert {"sense", "fuse", "plan"} <= set(times)
    # (template arguments as the launch site spells them: K3 <cells per lane, misaligned rows, explicit flips, sense records, dense lane
    #  mapping, area sums, tile storage[, wavefronts per workgroup, loads in flight per lane]>, the fusion <misaligned rows, area sums[, tile storage]>)
    tl = "true" if env.tiled else "false"
    for cls, kernel in (("sense", f"k_sense_tiles<4, false, false, true, true, false, {tl}"),
                        ("fuse", "k_fuse_tiles<false, false, true>" if env.tiled else "k_fuse_tiles<false, false>"), ("plan", "k_plan_step")):
        rec = times[cls]
        assert rec["launches"] == 5 and rec["kernel"].startswith(kernel) and rec["kernel"].endswith(">" if "<" in kernel else "p"), (cls, rec)
        assert 1.0 < rec["min_us"] <= rec["avg_us"] < 2000.0, (cls, rec)
    assert env.event_times_us() == {}                            # reading resets
    env.reset(np.arange(100, 164))                               # the reset's kernels have their own classes
    env.profile = True
    env.reset(np.arange(200, 264))
    env.profile = False
    times = env.event_times_us()
    assert times["reset_maps"]["launches"] == 1 and times["reset_maps"]["kernel"] == "k_reset_maps"
    assert times["reset"]["launches"] == 2 and times["reset"]["kernel"] == "k_fill_truth"   # scalars, then the half-plane truth (the last one names the class)
