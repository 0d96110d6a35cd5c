"""The oracle (oracle/ipp_oracle.py) against golden vectors captured from the reference itself
(oracle/make_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest

import ipp_oracle as O
from configs import make_params
from conftest import local_subset, unpack_correctness

RTOL = 1e-5  # BASELINE.json north_star: float posteriors/returns within 1e-5 relative; ints bit-exact


def lattice(d):
    return [np.array([x * d.spacing, y * d.spacing, d.min_altitude + z * d.spacing])
            for x in range(d.space_x) for y in range(d.space_y) for z in range(d.space_z)]


@pytest.mark.parametrize("name", ["default", "small", "c2", "c4", "c5"])
def test_derived_constants_and_footprints(golden, name):
    fx = golden("derived_footprints")
    d = O.Derived(make_params(name))
    assert [d.res_x, d.res_y] == list(fx[f"{name}_res"])  # bit-exact float64
    assert [d.gx, d.gy] == list(fx[f"{name}_dims"])
    assert [d.space_x, d.space_y, d.space_z] == list(fx[f"{name}_space"])
    full, clip, fixed = [], [], []
    for pos in lattice(d):
        f, c = O.project_field_of_view(d, pos)
        full.append(f), clip.append(c), fixed.append(O.fixed_footprint_coordinates(f, c))
    assert np.array_equal(np.array(full), fx[f"{name}_fp_full"])
    assert np.array_equal(np.array(clip), fx[f"{name}_fp_clip"])
    assert np.array_equal(np.array(fixed), fx[f"{name}_fp_fixed"])


def test_grid_sizes_match_survey(golden):
    assert [O.Derived(make_params(n)).gx for n in ("default", "small", "c2", "c4", "c5")] == [493, 128, 256, 512, 1024]


def test_start_states(golden):
    fx = golden("start_states")
    d = O.Derived(make_params("default"))
    got = np.array([[O.start_state(d, a, e) for a in range(16)] for e in range(1, 65)])
    assert np.array_equal(got, fx["seed3"])
    d7 = O.Derived(make_params("default", environment__seed=7))
    got7 = np.array([[O.start_state(d7, a, e) for a in range(4)] for e in range(1, 17)])
    assert np.array_equal(got7, fx["seed7"])
    assert list(got[0, 0]) == [25, 0, 15]  # agent 0 is always seeded with 0


def test_truth(golden):
    fx = golden("truth")
    got = np.array([O.truth_split_params(e) for e in range(1, 4097)])
    assert np.array_equal(got, fx["split_pct"])
    for key in [k for k in fx if k.startswith("field_")]:
        e = int(key.split("_")[1][1:])
        r, c = (int(v) for v in key.split("_")[2].split("x"))
        s, pc = O.truth_split_params(e)
        assert np.array_equal(O.truth_from_split(r, c, s, pc).astype(np.uint8), fx[key]), key


def test_random_field_terrain(golden):
    """The thresholded power-law field of ground_truths.py:25-40 (odd sizes leave one amplitude row/column at 0)."""
    fx = golden("terrain")
    for key in fx:
        e = int(key.split("_")[1][1:])
        r, c = (int(v) for v in key.split("_")[2].split("x"))
        want = np.unpackbits(fx[key])[: r * c].reshape(r, c)
        got = O.grf_field(r, c, e, 5.0).astype(np.uint8)
        assert np.array_equal(got, want), key
        assert 0.05 < got.mean() < 0.95


@pytest.mark.parametrize("A", [4, 6, 9, 27])
def test_action_masks_and_moves(golden, A):
    fx = golden("masks")
    over = dict(experiment__constraints__num_actions=A)
    if A in (4, 9):
        over.update(experiment__constraints__min_altitude=15, experiment__constraints__max_altitude=15)
    d = O.Derived(make_params("default", **over))
    got = np.array([O.action_mask(d, pos) for pos in fx[f"a{A}_pos"]])
    assert np.array_equal(got, fx[f"a{A}_mask"])
    base = np.array([25, 25, d.min_altitude + (5 if d.space_z > 1 else 0)])
    assert np.array_equal(np.array([O.action_to_position(d, base, a) for a in range(A)]), fx[f"a{A}_moves"])
    for pos, others, n, m_in, m_out in zip(fx[f"a{A}_col_pos"], fx[f"a{A}_col_others"], fx[f"a{A}_col_n"],
                                           fx[f"a{A}_col_in"], fx[f"a{A}_col_out"]):
        got = O.apply_collision_mask(d, pos, m_in.copy(), [others[k] for k in range(n)])
        assert np.array_equal(got, m_out), (pos, others[:n], m_in, m_out, got)


def test_communication(golden):
    fx = golden("comm")
    for rg, fail, pos, draws, rec in zip(fx["range"], fx["failure"], fx["pos"], fx["draws"], fx["received"]):
        for i in range(len(pos)):
            ks = O.received_set(pos, i, rg, fail, draws[i])
            assert ks == list(np.flatnonzero(rec[i])), (rg, fail, pos, i)
    d = O.Derived(make_params("default", experiment__uav__fix_range=False))
    assert [O.episode_comm_range(d, e) for e in range(1, 65)] == list(fx["episode_range"])


def test_bayes_update_and_measurement(golden):
    fx = golden("bayes_measurement")
    for prior in (0.5, 0.3):
        out = O.bayes_update(fx[f"p{prior}_x"].copy(), fx[f"p{prior}_y"], prior)
        np.testing.assert_allclose(out, fx[f"p{prior}_out"], rtol=1e-12)
        chain = np.full((6, 4), prior, dtype=np.float32)
        yv = np.float32(np.round(np.array([0.01, 0.99, 0.265, 0.735, 0.375, 0.625]), 3))[:, None] * np.ones((1, 4), dtype=np.float32)
        for k in range(20):
            chain = np.float32(O.bayes_update(chain, yv, prior))
            np.testing.assert_array_equal(chain, fx[f"p{prior}_chain"][k])
    truth = fx["meas_truth"].astype(np.float64)
    for alt in (5, 10, 15):
        m = O.noisy_measurement(truth, O.noise_of_altitude(alt), fx[f"meas_corr_{alt}"].astype(np.int64))
        assert m.dtype == np.float32
        np.testing.assert_array_equal(m, fx[f"meas_out_{alt}"])


def test_entropy_and_reward(golden):
    fx = golden("entropy_reward")
    d = O.Derived(make_params("small"))
    for k in range(3):
        after, before, truth = fx[f"after{k}"], fx[f"before{k}"], fx[f"truth{k}"].astype(np.float64)
        for mode in ("reward", "eval", "global"):
            wh, w, h, _, p = O.w_entropy_map(d, None, after.copy(), truth, mode)
            np.testing.assert_allclose(wh, fx[f"{mode}{k}_wH"], rtol=1e-12, atol=0)
            np.testing.assert_array_equal(w, fx[f"{mode}{k}_w"])
            np.testing.assert_allclose(h, fx[f"{mode}{k}_H"], rtol=1e-12)
            np.testing.assert_allclose(p, fx[f"{mode}{k}_p"], rtol=1e-12)
        _, rel, ab = O.global_reward(d, before.copy(), after.copy(), truth)
        np.testing.assert_allclose([rel, ab], fx[f"reward{k}"], rtol=1e-12)
        s1, s2 = O.reward_sums(d, before, after)
        np.testing.assert_allclose([22 * s1 / s2 - 0.5, 10 * s1 / before.size - 0.17], fx[f"reward{k}"], rtol=RTOL, atol=1e-7)


EPISODES = {
    "episode_c2_e1": dict(name="c2", over={}),
    "episode_small27_e6": dict(name="small", over=dict(experiment__missions__n_agents=3, experiment__uav__fix_range=False,
                                                       experiment__uav__failure_rate=0.3, experiment__constraints__num_actions=27)),
    "episode_small5_e3": dict(name="small", over=dict(experiment__missions__n_agents=5, experiment__uav__communication_range=15)),
    # the reference's default grid, 493 x 493 (11 feature bins that are not whole cells wide), 2 UAVs = BASELINE config 1's team
    "episode_default_e2": dict(name="c1", over={}),
    # mapping.prior = 0.3: every fused message shifts every cell of the grid (the fusion's explicit slow path), recorded from the reference
    "episode_small_prior03_e4": dict(name="small", over=dict(mapping__prior=0.3, experiment__missions__n_agents=3)),
    # BASELINE config 4's team and grid (8 UAVs, 512 x 512; plans of up to nine ops); final local maps of agents 0 and 5 only
    "episode_c4_e2": dict(name="c4", over={}),
}


def replay(fx, params):
    """Drive the oracle with the randomness the reference consumed in the recorded episode."""
    d = O.Derived(params)
    n, T = d.n_agents, d.budget + 1
    corr = unpack_correctness(fx)
    comm = fx["comm_draws"]
    assert len(comm) == T * n * n
    ep = O.OracleEpisode(
        params, int(fx["episode"]),
        correctness=lambda i, s, shape: corr[s * n + i].reshape(shape),
        choose_action=lambda i, t, mask, obs: fx["actions"][t, i],
        comm_draw=lambda i, j, t: comm[(t * n + i) * n + j])
    return d, ep, ep.run()


@pytest.mark.parametrize("tag", list(EPISODES))
def test_full_episode_replay(golden, tag):
    fx = golden(tag)
    params = make_params(EPISODES[tag]["name"], **EPISODES[tag]["over"])
    d, ep, log = replay(fx, params)
    assert np.array_equal(ep.truth.astype(np.uint8), fx["truth"])
    for t, rec in enumerate(log):
        assert np.array_equal(rec["positions"], fx["positions"][t]), t          # grid indices: bit-exact
        assert np.array_equal(rec["next_positions"], fx["positions"][t + 1]), t
        assert np.array_equal(rec["masks"], fx["masks"][t]), t                    # action masks: bit-exact
        np.testing.assert_allclose(rec["relative_reward"], fx["rewards"][t, 0], rtol=RTOL)
        np.testing.assert_allclose(np.array(rec["observations"]), fx["obs"][t], rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(np.array(rec["states"]), fx["state"][t], rtol=RTOL, atol=1e-7)
        assert rec["done"] == bool(fx["done"][t, 0])
    np.testing.assert_allclose(sum(r["relative_reward"] for r in log), fx["episode_return"], rtol=RTOL)
    np.testing.assert_allclose(sum(r["absolute_reward"] for r in log), fx["abs_return"], rtol=RTOL)
    np.testing.assert_allclose(local_subset(fx, np.array([a["local_map"] for a in ep.agents])), fx["final_local"], rtol=RTOL)
    np.testing.assert_allclose(ep.global_map, fx["final_global"], rtol=RTOL)
    if "global_t0" in fx:   # (the 493 x 493 fixture keeps only the final maps)
        np.testing.assert_allclose(log[0]["global_map"], fx["global_t0"], rtol=RTOL)
        np.testing.assert_allclose(log[7]["global_map"], fx["global_t7"], rtol=RTOL)


def test_td_lambda(golden):
    fx = golden("td_lambda")
    g, lam = float(fx["gamma"]), float(fx["lam"])
    for a in range(fx["rewards"].shape[0]):
        td, dr = O.td_lambda_targets(fx["rewards"][a], fx["dones"][a], fx["qsel"][a], g, lam)
        np.testing.assert_allclose(td, fx["td"][a], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dr, fx["dr"][a], rtol=1e-5, atol=1e-6)
        assert td[15] == 0.0 and td[30] == 0.0 and dr[15] == 0.0  # first step of later episodes
    td, dr = O.td_lambda_targets(fx["rewards"][0][:15], fx["dones"][0][:15], fx["qsel"][0][:15], g, lam)
    np.testing.assert_allclose(td, fx["td_single"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dr, fx["dr_single"], rtol=1e-5, atol=1e-6)


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    r = O.philox4x32(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    r = O.philox4x32(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(x) for x in r] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    r = O.philox4x32(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)
    assert [int(x) for x in r] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


@pytest.mark.parametrize("tag", list(EPISODES))
def test_exact_mode_differs_only_by_reference_quantisation(golden, tag):
    """The oracle's exact-float64 mode (what the GPU path is held to at 1e-5, every cell) against the reference's
    own output: identical integers, and floats that differ only by the reference's float32 map re-quantisation."""
    from conftest import REFERENCE_QUANTISATION_CELLS as RQ, assert_posteriors
    fx = golden(tag)
    params = make_params(EPISODES[tag]["name"], **EPISODES[tag]["over"])
    d = O.Derived(params)
    n = d.n_agents
    corr = unpack_correctness(fx)
    comm = fx["comm_draws"]
    ep = O.OracleEpisode(params, int(fx["episode"]), correctness=lambda i, s, shape: corr[s * n + i].reshape(shape),
                         choose_action=lambda i, t, mask, obs: fx["actions"][t, i],
                         comm_draw=lambda i, j, t: comm[(t * n + i) * n + j], exact=True)
    log = ep.run()
    # (mapping.prior != 0.5: every fused message shifts EVERY cell of the grid, so the reference's float32 re-quantisation touches
    #  every cell at every fusion: its S1 / S2 carry 1e-6 of relative noise, 22 times that in the reward, and its area averages 1e-5)
    shifted = d.prior != 0.5
    for t, rec in enumerate(log):
        assert np.array_equal(rec["next_positions"], fx["positions"][t + 1])
        assert np.array_equal(rec["masks"], fx["masks"][t])
        np.testing.assert_allclose(rec["relative_reward"], fx["rewards"][t, 0], rtol=RTOL, atol=2.5e-5 if shifted else 1e-6)
        np.testing.assert_allclose(np.array(rec["observations"]), fx["obs"][t], rtol=RTOL, atol=1e-5 if shifted else 1e-6)
    assert_posteriors(local_subset(fx, np.array([a["local_map"] for a in ep.agents])), fx["final_local"], strict=False, msg="final local", allow=RQ.get((tag, "final_local"), []))
    assert_posteriors(ep.global_map, fx["final_global"], strict=False, msg="final global", allow=RQ.get((tag, "final_global"), []))
    # ... and the list is DERIVED, not tuned: the cells listed for this recording are exactly the cells in which the
    # exact-float64 oracle and the reference's own recording differ by more than 1e-5 -- no cell is listed that does not need it, so
    # the looser 3e-4 cap of assert_posteriors applies nowhere else.  (The GPU path itself is held to 1e-5 in EVERY cell against
    # this exact-mode oracle; the list only ever loosens the comparison with the reference's float32-re-quantised recording.)
    for key, got, want in (("final_local", local_subset(fx, np.array([a["local_map"] for a in ep.agents])), fx["final_local"]),
                           ("final_global", ep.global_map, fx["final_global"])):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(got == want, 0.0, np.abs(got - want) / np.abs(want))
        deviating = sorted(tuple(int(v) for v in idx) for idx in np.argwhere(rel > 1e-5))
        assert deviating == sorted(tuple(c) for c in RQ.get((tag, key), [])), (tag, key, deviating)


IG_CASES = {"ig_c1_e1": dict(name="c1", over={}), "ig_small3_e4": dict(name="small", over=dict(experiment__missions__n_agents=3))}


def oracle_ig_run(fx, tag):
    """The oracle's IG_baseline rerun from the draws the reference recorded: (execute()'s dict, the decidable F1 counts per evaluation)."""
    params = make_params(IG_CASES[tag]["name"], **IG_CASES[tag]["over"])
    n = params["experiment"]["missions"]["n_agents"]
    corr = unpack_correctness(fx)
    comm = fx["comm_draws"]

    def correctness(i, s, shape):
        # the reference draws in call order: n start sensings, then n per step
        return corr[s * n + i].reshape(shape)

    ig = O.OracleIGBaseline(params, int(fx["episode"]), correctness, comm_draw=lambda i, j, t: comm[(t * n + i) * n + j])
    with O.record_f1_counts() as counts:
        out = ig.execute()
    return out, counts


@pytest.mark.parametrize("tag", list(IG_CASES))
def test_ig_baseline_replay(golden, tag):
    """BASELINE config 1 (2 UAVs, default 493 x 493 grid, greedy information-gain planner on the CPU) and a smaller case:
    the oracle's IG_baseline restatement against the reference's own run (SURVEY Q18 known answers)."""
    fx = golden(tag)
    out, counts = oracle_ig_run(fx, tag)
    _check_counts_bracket_recorded_f1(counts, fx["f1"])
    assert np.array_equal(np.array(out["altitudes"]), fx["altitudes"])
    np.testing.assert_allclose(np.array(out["gains"]), fx["gains"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(out["entropies"], fx["entropies"], rtol=RTOL)
    np.testing.assert_allclose(out["f1"], fx["f1"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(out["relative_return"], fx["relative_return"], rtol=RTOL)
    np.testing.assert_allclose(out["absolute_return"], fx["absolute_return"], rtol=RTOL)
    if tag == "ig_c1_e1":   # the baseline's returns are constants, the informative outputs are entropy and F1
        np.testing.assert_allclose(out["relative_return"], -7.5, rtol=1e-9)
        np.testing.assert_allclose(out["absolute_return"], -2.55, rtol=1e-9)
        assert abs(out["entropies"][0] - 1.0) < 1e-12 and out["f1"][0] == 0.0
        assert 0.55 < out["entropies"][-1] < 0.70 and 0.65 < out["f1"][-1] < 0.80   # sensor-noise dependent (survey probe: 0.618 / 0.729)
        assert [a[0] for a in out["altitudes"]][:3] == [15, 10, 10]                 # descends after the first step


def _unpack(fx):
    from conftest import unpack_correctness
    return unpack_correctness(fx)


def _check_counts_bracket_recorded_f1(counts, recorded):
    """The decidable counts (thresholds +-1e-5 in log-odds) bracket the F1 the reference recorded: strict tp with lax fp is the
    lowest attainable score, lax tp with strict fp the highest; and tp + fn is the number of target cells under either."""
    assert len(counts) == len(recorded)
    for (s, l), want in zip(counts, recorded):
        assert s[0] + s[2] == l[0] + l[2] and s[0] <= l[0] and s[1] <= l[1]
        lo = 2 * s[0] / max(2 * s[0] + l[1] + s[2], 1)
        hi = 2 * l[0] / max(2 * l[0] + s[1] + l[2], 1)
        assert lo - 1e-12 <= want <= hi + 1e-12, (lo, want, hi)


def oracle_random_run(fx):
    """random_baseline.py rerun by the oracle from the recorded draws: (entropies, f1s, decidable F1 counts per evaluation)."""
    params = make_params("small", experiment__missions__n_agents=3)
    d = O.Derived(params)
    n, corr, ep = 3, _unpack(fx), int(fx["episode"])
    truth = O.make_truth(d, ep)
    pos = [O.start_state(d, i, ep) for i in range(n)]
    visits = [[p.copy() for p in pos]]
    for t in range(1, d.budget + 1):
        for i in range(n):
            a = int(fx["actions"][(t - 1) * n + i])
            assert O.action_mask(d, pos[i])[a] == 1
            pos[i] = O.action_to_position(d, pos[i], a)
        visits.append([p.copy() for p in pos])
    with O.record_f1_counts() as counts:
        ent, f1 = O.shared_map_curves(d, truth, visits, lambda k: corr[k])
    return ent, f1, counts


def test_random_baseline_curves(golden):
    """random_baseline.py rerun by the oracle from the recorded draws (start cells from the legacy seed rule)."""
    fx = golden("random_small3_e6")
    ent, f1, counts = oracle_random_run(fx)
    np.testing.assert_allclose(ent, fx["entropies"], rtol=1e-6)
    np.testing.assert_allclose(f1, fx["f1"], rtol=1e-9)   # incl. the exactly-cancelled cells (same float32 rounding noise)
    _check_counts_bracket_recorded_f1(counts, fx["f1"])


def oracle_lawnmower_run(fx):
    params = make_params("small", experiment__missions__n_agents=8, experiment__baselines__lawnmower__altitude=10)
    d = O.Derived(params)
    corr = _unpack(fx)
    paths = O.lawnmower_paths(10)
    visits = [[p[idx] for p in paths] for idx in range(15)]
    with O.record_f1_counts() as counts:
        ent, f1 = O.shared_map_curves(d, O.make_truth(d, int(fx["episode"])), visits, lambda k: corr[k])
    return ent, f1, counts


def test_lawn_mower_curves(golden):
    fx = golden("lawnmower_small_e2")
    ent, f1, counts = oracle_lawnmower_run(fx)
    np.testing.assert_allclose(ent, fx["entropies"], rtol=1e-6)
    np.testing.assert_allclose(f1, fx["f1"], rtol=1e-9)
    _check_counts_bracket_recorded_f1(counts, fx["f1"])


def oracle_comatest_run(fx):
    params = make_params("small", experiment__missions__n_agents=3)
    n, corr = 3, _unpack(fx)
    with O.record_f1_counts() as counts:
        ent, f1, stages = O.deployment_curves(params, int(fx["episode"]), lambda t, i: int(fx["actions"][t * n + i]),
                                              lambda s, i: corr[s * n + i])
    return ent, f1, stages, counts


def test_coma_test_curves(golden):
    fx = golden("comatest_small3_e9")
    ent, f1, stages, counts = oracle_comatest_run(fx)
    _check_counts_bracket_recorded_f1(counts, fx["f1"])
    assert np.array_equal(stages, fx["positions"])
    np.testing.assert_allclose(ent, fx["entropies"], rtol=1e-6)
    np.testing.assert_allclose(f1, fx["f1"], rtol=1e-9)
    assert np.isclose(float(fx["ret"]), -0.17 * 15) and np.isclose(float(fx["relative_return"]), -0.5 * 15)
