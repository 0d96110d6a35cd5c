"""Tile storage of the belief maps (include/ippmarl.h: ippm_set_map_layout; DESIGN.md "tile storage").

The layout changes WHERE a cell lives, never what it holds: an env whose maps are stored as 128-byte tiles of 4 rows x 8 cells must fly
the same episodes, bit for bit, as the row-major env -- whose parity with the oracle and the reference's recordings the rest of the suite
holds (and the whole GPU suite runs in either layout: IPPM_MAP_TILED=0 / 1 decide what map_layout="auto" means)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from configs import make_params  # noqa: E402


def _cell_index(x, y, gy):
    """include/ippmarl.h, ippm_set_map_layout: float index of cell (x, y) in tile storage."""
    return (x >> 2) * 4 * gy + (y >> 3) * 32 + (x & 3) * 8 + (y & 7)


@pytest.mark.parametrize("gx,gy", [(4, 8), (12, 24), (256, 256), (8, 1024)])
def test_rows_view_is_the_documented_index_map(gx, gy):
    from ippmarl.vec_env import rows_view, tiles_view
    rows = torch.arange(2 * 3 * gx * gy, dtype=torch.float32).reshape(2, 3, gx, gy)
    stored = tiles_view(rows, True)
    assert stored.shape == rows.shape
    assert torch.equal(rows_view(stored, True), rows)
    assert rows_view(rows, False) is rows and tiles_view(rows, False) is rows
    flat = stored.reshape(2, 3, -1)
    rng = np.random.RandomState(gx * 1000 + gy)
    for _ in range(200):
        x, y = int(rng.randint(gx)), int(rng.randint(gy))
        assert flat[1, 2, _cell_index(x, y, gy)] == rows[1, 2, x, y]
    # a tile is one 128-byte line: 32 consecutive floats hold rows 4R..4R+3 x columns 8C..8C+7
    t = flat[0, 0, :32].reshape(4, 8)
    assert torch.equal(t, rows[0, 0, :4, :8])


def _both(params, n_envs, **kw):
    from ippmarl.vec_env import VecEnv
    return VecEnv(params, n_envs, map_layout="rows", **kw), VecEnv(params, n_envs, map_layout="tiles", **kw)


@pytest.mark.gpu
def test_relayout_entry_point_and_layout_query():
    from ippmarl import _ffi
    from ippmarl.vec_env import VecEnv, rows_view, tiles_view
    params = make_params("small")
    env = VecEnv(params, 2, map_layout="tiles", track_area=False)
    yes = np.zeros(1, dtype=np.int32)
    env.ctx.call("ippm_map_layout", yes.ctypes.data)
    assert env.tiled and yes[0] == 1
    rows = torch.randn(5, env.d.grid_x, env.d.grid_y, device=env.device)
    stored = torch.empty_like(rows)
    back = torch.empty_like(rows)
    env.ctx.call("ippm_maps_relayout", _ffi.ptr(rows), _ffi.ptr(stored), 5, 1, env.stream)
    env.ctx.call("ippm_maps_relayout", _ffi.ptr(stored), _ffi.ptr(back), 5, 0, env.stream)
    torch.cuda.synchronize()
    assert torch.equal(stored, tiles_view(rows, True).contiguous()) and torch.equal(back, rows)
    assert torch.equal(rows_view(stored, True), rows)
    # a grid that is not made of whole tiles (the reference's default 493 x 493) keeps row-major maps: "auto" falls back, "tiles" refuses
    dflt = make_params("default")
    assert not VecEnv(dflt, 1, map_layout="auto", track_area=False).tiled
    with pytest.raises(_ffi.IppmError):
        VecEnv(dflt, 1, map_layout="tiles", track_area=False)


@pytest.mark.gpu
def test_layout_advice_goes_by_the_batch():
    """ippm_map_layout_advice (what map_layout="auto" follows): tiles when the batch's maps take 2 GB or more and footprint rows are at most 256
    cells -- BASELINE config 2 as quoted (1024 envs: 1.3 GB) stays row-major, the same shape at 2048 envs and config 4's per-GPU shape take tiles,
    config 5's 1024 x 1024 grid (rows of up to 360 cells) and grids that are not made of whole tiles never do."""
    from ippmarl.vec_env import VecEnv

    def advice(name, n_envs, **over):
        env = VecEnv(make_params(name, **over), 1, map_layout="rows", track_area=False)
        out = np.zeros(1, dtype=np.int32)
        env.ctx.call("ippm_map_layout_advice", n_envs, out.ctypes.data)
        return int(out[0])
    assert advice("c2", 1024) == 0 and advice("c2", 2048) == 1 and advice("c2", 4096) == 1
    assert advice("c2", 1024, experiment__missions__n_agents=8) == 1
    assert advice("c4", 1024) == 1 and advice("c4", 128) == 0
    assert advice("c5", 64) == 0 and advice("c5", 4096) == 0
    assert advice("default", 100000) == 0
    # ... and the sub-batches of a split batch are asked about the WHOLE batch
    from ippmarl.vec_env import SplitVecEnv
    old = os.environ.pop("IPPM_MAP_TILED", None)
    try:
        small = SplitVecEnv(make_params("c4"), 8, parts=2)
        assert not small.tiled and not VecEnv(make_params("c2"), 16, track_area=False).tiled
        # ... and an env with tracked area sums (the training rollout) keeps rows whatever the batch: its LDS atomics are slower on tiles
        assert VecEnv(make_params("c4"), 2, track_area=False, layout_envs=1024).tiled
        assert not VecEnv(make_params("c4"), 2, track_area=True, layout_envs=1024).tiled
    finally:
        if old is not None:
            os.environ["IPPM_MAP_TILED"] = old


CASES = {
    "c2": ("c2", {}, 6),
    "c2_9actions_failures": ("c2", {"experiment__constraints__num_actions": 9, "experiment__uav__failure_rate": 0.2,
                                    "experiment__uav__communication_range": 40}, 5),
    "small_6uavs_range": ("small", {"experiment__missions__n_agents": 6, "experiment__uav__fix_range": False}, 9),
    "c4": ("c4", {}, 3),
    "c5_5uavs": ("c5", {"experiment__missions__n_agents": 5}, 2),
}


@pytest.mark.gpu
@pytest.mark.parametrize("track", [False, True])
@pytest.mark.parametrize("case", list(CASES))
def test_tiled_env_flies_the_same_episodes_bit_for_bit(case, track):
    """Two waves of whole episodes under the uniform random policy, random-field terrain: positions, actions, comm matrices and every
    cell of every map equal the row-major env's after every step; rewards and (tracked form) the network inputs to summation order
    (float64 atomics whose order the launch decides -- in either layout)."""
    from ippmarl.vec_env import POLICY_UNIFORM
    name, over, n_envs = CASES[case]
    params = make_params(name, **over)
    a, b = _both(params, n_envs, track_area=track, terrain="random_field")
    assert not a.tiled and b.tiled
    for wave in range(2):
        eps = np.arange(1, n_envs + 1) * 13 + 700 * wave
        a.reset(eps)
        b.reset(eps)
        assert torch.equal(a.truth, b.truth) and torch.equal(a.pos, b.pos)
        assert torch.equal(b.rows_view(b.local), a.local), (wave, "reset")
        for t in range(a.d.budget + 1):
            oa = a.build_observations(t, features=True) if track else None
            ob = b.build_observations(t, features=True) if track else None
            ra, _, sa = a.steps(t, policy=POLICY_UNIFORM, features=track)
            rb, _, sb = b.steps(t, policy=POLICY_UNIFORM, features=track)
            assert torch.equal(a.pos, b.pos) and torch.equal(a.action, b.action) and torch.equal(a.comm, b.comm), (wave, t)
            assert torch.equal(b.rows_view(b.local), a.local), (wave, t)
            assert torch.equal(b.rows_view(b.glob), a.glob), (wave, t)
            assert torch.equal(a.code, b.code) and torch.equal(a.rect, b.rect), (wave, t)
            torch.testing.assert_close(rb, ra, rtol=1e-6, atol=1e-6)
            if track:
                torch.testing.assert_close(ob, oa, rtol=1e-6, atol=1e-6)
                torch.testing.assert_close(sb, sa, rtol=1e-6, atol=1e-6)
        assert int(a.fault.abs().sum()) == 0 and int(b.fault.abs().sum()) == 0
    ca, cb = a.counters(reset=True), b.counters(reset=True)
    for k in ("sense_cells", "fuse_local_ops", "fuse_global_ops", "fuse_local_cells", "fuse_global_cells"):
        assert ca[k] == cb[k], (k, ca[k], cb[k])
    # exports agree: probabilities, evaluation metrics, planner gains
    assert torch.equal(b.posterior_local(), a.posterior_local()) and torch.equal(b.posterior_global(), a.posterior_global())


@pytest.mark.gpu
def test_tiled_env_single_purpose_entry_points_and_metrics():
    """Everything else that takes maps reads the context's layout: the row walker behind ippm_fuse_global_reward, the streaming area
    sums, the full-grid weighted entropy and F1 counts against the truth bits, the information-gain planner's candidate walk."""
    from ippmarl import _ffi
    from ippmarl.vec_env import POLICY_UNIFORM
    params = make_params("c2", experiment__constraints__num_actions=9)
    a, b = _both(params, 4, track_area=True, terrain="random_field")
    eps = np.array([5, 6, 7, 8])
    a.reset(eps)
    b.reset(eps)
    for t in range(6):
        for env in (a, b):
            env.build_observations(t, features=True)
            env.steps(t, policy=POLICY_UNIFORM, features=True)
    assert torch.equal(b.rows_view(b.local), a.local)
    out = []
    for env in (a, b):
        E, N = env.E, env.d.n_agents
        ent = torch.zeros(E, dtype=torch.float64, device=env.device)
        env.ctx.call("ippm_weighted_entropy", env._p(env.glob), env._p(env.truth), 1, _ffi.ptr(ent), E, env.stream)
        counts = torch.zeros(E, 3, dtype=torch.int64, device=env.device)
        env.ctx.call("ippm_f1_counts", env._p(env.glob), env._p(env.truth), 1, 0.0, _ffi.ptr(counts), E, env.stream)
        fresh = torch.zeros_like(env.area)
        env.ctx.call("ippm_area_sums", env._p(env.local), _ffi.ptr(fresh), E * N, N, 0, env.stream)
        env.ctx.call("ippm_area_sums", env._p(env.glob), _ffi.ptr(fresh), E, 1, N, env.stream)
        acts = env.ig_actions(communication=True).clone()
        gains = env.ig_gains.clone()
        # the fresh measurements fused into a copy of the global maps by the stand-alone entry point (row walker)
        glob, ws, sums = env.glob.clone(), env.ws.clone(), env.sums.clone()
        reward = torch.empty_like(env.reward)
        env.ctx.call("ippm_fuse_global_reward", _ffi.ptr(glob), env._p(env.code), env._p(env.rect), env._p(env.pos), _ffi.ptr(ws),
                     _ffi.ptr(sums), _ffi.ptr(reward), E, env.stream)
        torch.cuda.synchronize()
        out.append((ent, counts, fresh, acts, gains, env.rows_view(glob), reward))
    (ea, ca, fa, aa, ga, ma, ra), (eb, cb, fb, ab, gb, mb, rb) = out
    assert torch.equal(ca, cb) and torch.equal(aa, ab) and torch.equal(ma, mb)
    torch.testing.assert_close(eb, ea, rtol=1e-6, atol=0)     # (float32 block sums in storage order)
    torch.testing.assert_close(fb, fa, rtol=1e-12, atol=1e-9)
    torch.testing.assert_close(gb, ga, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(rb, ra, rtol=1e-6, atol=1e-6)
