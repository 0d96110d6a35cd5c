"""CPU oracle for the ipp-marl hot path (env step + COMA feature/target arithmetic).

TEST INFRASTRUCTURE ONLY.  Nothing under ``ipp-marl_amd/`` may import this file.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use
it, and only as the checker / the timed CPU baseline -- never as the product path.

It is a single-environment NumPy restatement of what dmar-bonn/ipp-marl computes for
one multi-UAV episode, written from the behaviour of the reference (file:line cited per
function, paths relative to ``/root/reference/marl_framework/``).  It is pinned against
golden vectors captured by importing the reference in the build container
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).

Two third-party pieces of arithmetic are *not* under /root/reference and are restated
from their published definition (parity unpinned at those two boundaries):
  * ``cv2.resize(..., INTER_AREA)`` (opencv-python 4.5.5.62) -> :func:`area_resize`
    (exact area-weighted average);
  * ``torch.multinomial`` streams (torch 1.13) -> randomness is an explicit *input* of
    every oracle function (correctness masks, chosen actions, comm draws), or comes from
    the counter-based Philox4x32-10 defined here and mirrored bit-for-bit on the device.
"""
from __future__ import annotations

import copy
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# derived constants (mapping/grid_maps.py:17-66, agent/state_space.py:10-21)
# --------------------------------------------------------------------------------------


class Derived:
    """Constants the reference re-derives from ``params`` on every call."""

    def __init__(self, params: Dict):
        env = params["environment"]
        con = params["experiment"]["constraints"]
        fov = params["sensor"]["field_of_view"]
        pix = params["sensor"]["pixel"]
        self.params = params
        self.seed = env["seed"]
        self.x_dim_m = env["x_dim"]
        self.y_dim_m = env["y_dim"]
        self.spacing = con["spacing"]
        self.min_altitude = con["min_altitude"]
        self.max_altitude = con["max_altitude"]
        self.budget = con["budget"]
        self.num_actions = con["num_actions"]
        self.angle_x = fov["angle_x"]
        self.angle_y = fov["angle_y"]
        # mapping/grid_maps.py:53-66
        self.res_x = (2 * self.min_altitude * math.tan(math.radians(self.angle_x) * 0.5)) / pix["number_x"]
        self.res_y = (2 * self.min_altitude * math.tan(math.radians(self.angle_y) * 0.5)) / pix["number_y"]
        # mapping/grid_maps.py:29-32,47-50 (int() truncation)
        self.gx = int(self.x_dim_m / self.res_x)
        self.gy = int(self.y_dim_m / self.res_y)
        # agent/state_space.py:16-21
        self.space_x = self.x_dim_m // self.spacing + 1
        self.space_y = self.y_dim_m // self.spacing + 1
        self.space_z = (self.max_altitude - self.min_altitude) // self.spacing + 1
        self.prior = params["mapping"]["prior"]
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        uav = params["experiment"]["uav"]
        self.comm_range = uav["communication_range"]
        self.fix_range = uav["fix_range"]
        self.failure_rate = uav["failure_rate"]
        # exact=False reproduces the reference's dtype flow bit for bit, including its float32 re-quantisation of
        # every map at the start of each fusion (mappings.py:83,93,100: np.float32(own_map_state.copy())).
        # exact=True keeps the maps in float64 throughout: the mathematically exact recursion, free of that
        # storage noise (see tests/conftest.py::assert_posteriors for how the two are used).
        self.exact = False


def position_to_index(d: Derived, position) -> np.ndarray:
    """agent/state_space.py:53-57"""
    return np.array([position[0] // d.spacing, position[1] // d.spacing, position[2] // d.spacing - 1])


def start_state(d: Derived, agent_id: int, episode: int) -> np.ndarray:
    """agent/state_space.py:28-51 -- legacy MT19937 stream seeded seed*episode*agent_id, z fixed 15."""
    r = np.random.RandomState(seed=d.seed * episode * agent_id)
    sx = d.spacing * r.randint(0, d.space_x)
    sy = d.spacing * r.randint(0, d.space_y)
    return np.array([sx, sy, 15])


# --------------------------------------------------------------------------------------
# sensor model, footprint projection (sensors/models/sensor_models.py:13-22, sensors/cameras.py:31-79)
# --------------------------------------------------------------------------------------


def noise_of_altitude(altitude) -> float:
    if altitude == 5:
        return 0.01
    if altitude == 10:
        return 0.265
    if altitude == 15:
        return 0.375
    return 0


def project_field_of_view(d: Derived, position) -> Tuple[List[int], List[int]]:
    """Returns ([yu,yd,xl,xr] unclipped, same clipped to [0,dim-1]); both axes use res_x for the centre
    (sensors/cameras.py:62-77).  Expression order is kept: the floor() sits on a float64 knife edge."""
    position = np.asarray(position)
    x_range_m = 2 * position[2] * np.tan(0.5 * np.radians(d.angle_x))
    y_range_m = 2 * position[2] * np.tan(0.5 * np.radians(d.angle_y))
    x_cells = np.floor(x_range_m / d.res_x)
    y_cells = np.floor(y_range_m / d.res_y)
    centre = np.floor(position[:2] / d.res_x)
    radius = np.floor(0.5 * np.array([x_cells, y_cells]))
    xl, yu = centre - radius
    xr, yd = centre + radius
    full = [int(yu), int(yd), int(xl), int(xr)]
    xl, xr = np.clip(np.array([xl, xr]), 0, d.gx - 1)
    yu, yd = np.clip(np.array([yu, yd]), 0, d.gy - 1)
    return full, [int(yu), int(yd), int(xl), int(xr)]


def fixed_footprint_coordinates(fp, fc) -> Tuple[int, int, int, int]:
    """Where the clipped tile sits inside the unclipped-size image (utils/utils.py:79-98)."""
    h, w = fp[1] - fp[0], fp[3] - fp[2]
    yu, yd, xl, xr = 0, h, 0, w
    if fc[0] > fp[0]:
        yu = h - (fc[1] - fc[0])
    if fc[1] < fp[1]:
        yd = fc[1] - fc[0]
    if fc[3] < fp[3]:
        xr = fc[3] - fc[2]
    if fc[2] > fp[2]:
        xl = w - (fc[3] - fc[2])
    return int(yu), int(yd), int(xl), int(xr)


# --------------------------------------------------------------------------------------
# ground truth (mapping/ground_truths.py:16-56; the FFT field at :25-40 is computed and discarded)
# --------------------------------------------------------------------------------------


def truth_split_params(episode: int) -> Tuple[int, int]:
    """(split_idx, percentage) drawn as ground_truths.py:43-48 does, from np.random.seed(episode)."""
    rs = np.random.RandomState(episode)  # same legacy stream as np.random.seed(episode)
    split = int(rs.randint(4))
    pct = int(rs.randint(30, 61))
    return split, pct


def truth_from_split(rows: int, cols: int, split: int, pct: int) -> np.ndarray:
    """Half-plane split field, float64 in {0,1} (ground_truths.py:42-56).  rows=y_dim, cols=x_dim."""
    field = np.zeros((rows, cols))
    if split == 0:
        field[: int((rows * pct) / 100), :] = 1
    elif split == 1:
        field[int((rows * (1 - pct)) / 100):, :] = 1
    elif split == 2:
        field[:, : int((cols * pct) / 100)] = 1
    elif split == 3:
        field[:, int((cols * (1 - pct)) / 100):] = 1
    return field


def make_truth(d: Derived, episode: int) -> np.ndarray:
    """Simulation.simulate_map (mapping/simulations.py:34-40): gaussian_random_field(pk, y_dim_pixel, x_dim_pixel)
    is called with (x_dim=self.y_dim_pixel, y_dim=self.x_dim_pixel) -> array of shape (gx, gy)."""
    split, pct = truth_split_params(episode)
    return truth_from_split(d.gx, d.gy, split, pct)


def grf_field(rows: int, cols: int, episode: int, exponent: float) -> np.ndarray:
    """The thresholded power-law random field the reference computes and throws away
    (ground_truths.py:19-40), vectorised.  Only used as *synthetic terrain* for benchmarks."""
    def idx(n):
        a = list(range(0, n // 2 + 1))
        b = [-i for i in reversed(range(1, n // 2))]
        return np.array(a + b, dtype=np.float64)

    rs = np.random.RandomState(episode)
    noise = np.fft.fft2(rs.normal(size=(rows, cols)))
    iy, ix = idx(rows), idx(cols)  # one index short for odd n: the reference leaves that amplitude row/col at 0
    ky, kx = np.meshgrid(iy, ix, indexing="ij")
    k = np.sqrt(ky ** 2 + kx ** 2)
    sub = np.zeros_like(k)
    nz = k > 0
    sub[nz] = np.sqrt(k[nz] ** (-exponent))
    amp = np.zeros((rows, cols))
    amp[: len(iy), : len(ix)] = sub
    f = np.fft.ifft2(noise * amp).real
    f = (f - f.min()) / (f.max() - f.min())
    return (f >= 0.5).astype(np.float64)


# --------------------------------------------------------------------------------------
# measurement + Bayesian occupancy update (mapping/simulations.py:42-65, mapping/mappings.py:32-132)
# --------------------------------------------------------------------------------------


def noisy_measurement(truth_tile: np.ndarray, noise: float, correctness: np.ndarray) -> np.ndarray:
    """correctness==1: cell observed correctly, 0: flipped (the reference draws it with torch.multinomial
    ([noise, 1-noise]); here it is an input).  Returns float32 in {noise, 1-noise} rounded to 3 decimals."""
    acc = 1 - noise
    g = truth_tile.copy()
    g = np.where(correctness == 0, abs(g - 1), g)
    g = acc * g
    np.putmask(g, (1 - acc) > g, 1 - acc)
    return np.float32(np.round(g, 3))


def bayes_update(x: np.ndarray, y: np.ndarray, prior: float) -> np.ndarray:
    """mappings.py:109-124.  Clips ``x`` IN PLACE (input only; the result is not clipped)."""
    x[0.9999 < x] = 0.9999
    x[0.0001 > x] = 0.0001
    l_xy = np.log(x / (1 - x)) + np.log(y / (1 - y))
    l_p = np.log(prior / (1 - prior))
    return 1 - (1 / (1 + np.exp(l_xy - l_p)))


def init_prior_map(d: Derived) -> np.ndarray:
    return np.full((int(d.gx), int(d.gy)), d.prior, dtype="float64" if d.exact else "float32")


def update_grid_map(d: Derived, truth: np.ndarray, position, map_state: np.ndarray, correctness: np.ndarray):
    """Sense at ``position`` and fuse into ``map_state`` in place (mappings.py:32-78).

    ``correctness`` has the clipped-tile shape.  Returns (map_state, cell_update, footprint_clipped,
    map2communicate, footprint_img) like the reference."""
    fp, fc = project_field_of_view(d, position)
    footprint_img = np.ones((fp[1] - fp[0], fp[3] - fp[2])) * 0.5
    section = map_state[fc[2]:fc[3], fc[0]:fc[1]]
    tile = truth[fc[2]:fc[3], fc[0]:fc[1]].copy()
    meas = noisy_measurement(tile, noise_of_altitude(position[2]), correctness)
    cell_update = bayes_update(section, meas, d.prior)
    map_state[fc[2]:fc[3], fc[0]:fc[1]] = cell_update
    m2c = np.ones_like(map_state) * 0.5
    m2c[fc[2]:fc[3], fc[0]:fc[1]] = meas
    ff = fixed_footprint_coordinates(fp, fc)
    footprint_img[ff[2]:ff[3], ff[0]:ff[1]] = meas
    return map_state, cell_update, fc, m2c, footprint_img


def tile_shape(fc) -> Tuple[int, int]:
    return (fc[3] - fc[2], fc[1] - fc[0])


def fuse_map(d: Derived, own: np.ndarray, others, agent_id, fusion_mode: str) -> np.ndarray:
    """mappings.py:80-104: sequential FULL-GRID updates with each other agent's map2communicate
    (0.5 outside its footprint => logit 0, but the input clip still applies to every cell)."""
    fused = own.astype(np.float64) if d.exact else np.float32(own.copy())
    if fusion_mode == "local":
        for key in others:
            if key == agent_id:
                continue
            fused = bayes_update(fused, np.float32(others[key]["map2communicate"]), d.prior)
    else:
        if isinstance(others, dict):
            for key in others:
                fused = bayes_update(fused, np.float32(others[key]["map2communicate"]), d.prior)
        else:
            for other in others:
                fused = bayes_update(fused, other, d.prior)
    return fused


# --------------------------------------------------------------------------------------
# action space (agent/action_space.py:25-589)
# --------------------------------------------------------------------------------------


def action_offsets(num_actions: int, spacing: int) -> np.ndarray:
    """[A,3] metre offsets (action_space.py:198-303)."""
    s = spacing
    if num_actions == 4:
        t = [[-s, 0, 0], [0, -s, 0], [0, s, 0], [s, 0, 0]]
    elif num_actions == 6:
        t = [[0, 0, s], [-s, 0, 0], [0, -s, 0], [0, s, 0], [s, 0, 0], [0, 0, -s]]
    elif num_actions == 9:
        t = [[dx * s, dy * s, 0] for dx in (-1, 0, 1) for dy in (-1, 0, 1)]
    elif num_actions == 27:
        t = [[dx * s, dy * s, dz * s] for dz in (1, 0, -1) for dx in (-1, 0, 1) for dy in (-1, 0, 1)]
    else:
        raise ValueError(num_actions)
    return np.array(t, dtype=np.int64)


def action_to_position(d: Derived, position, action: int) -> np.ndarray:
    return np.asarray(position) + action_offsets(d.num_actions, d.spacing)[int(action)]


def action_mask(d: Derived, position) -> np.ndarray:
    """Boundary mask (action_space.py:25-196), float64[A].

    Restated as "action a is valid iff the lattice cell it leads to exists", plus the variant quirks:
    9/27 actions forbid standing still; for 4/9 actions altitude never changes."""
    A = d.num_actions
    off = action_offsets(A, d.spacing)
    mask = np.ones(A)
    for a in range(A):
        nx, ny, nz = np.asarray(position) + off[a]
        ok = 0 <= nx <= d.x_dim_m and 0 <= ny <= d.y_dim_m
        if A in (6, 27):
            ok = ok and d.min_altitude <= nz <= d.max_altitude
        if A in (9, 27) and not off[a].any():
            ok = False
        mask[a] = 1.0 if ok else 0.0
    return mask


def collision_rule(num_actions: int, rel) -> List[int]:
    """Actions zeroed when an already-moved agent sits at lattice offset ``rel`` (action_space.py:309-589)."""
    dx, dy, dz = int(rel[0]), int(rel[1]), int(rel[2])
    if num_actions == 4:
        return {(-1, 0): [0], (0, -1): [1], (0, 1): [2], (1, 0): [3]}.get((dx, dy), [])
    if num_actions == 6:
        return {(0, 0): [0, 5], (-1, 0): [1], (0, -1): [2], (0, 1): [3], (1, 0): [4]}.get((dx, dy), [])
    if num_actions == 9:
        if abs(dx) > 1 or abs(dy) > 1 or (dx == 0 and dy == 0):
            return []
        return [(dx + 1) * 3 + (dy + 1)]
    if num_actions == 27:
        if abs(dx) > 1 or abs(dy) > 1 or abs(dz) > 1 or (dx == 0 and dy == 0 and dz == 0):
            return []
        c = (dx + 1) * 3 + (dy + 1)
        if dx == 0 and dy == 0:
            return [4, 22]
        return [c, c + 9, c + 18]
    raise ValueError(num_actions)


def apply_collision_mask(d: Derived, position, mask: np.ndarray, moved_positions) -> np.ndarray:
    """Order-dependent: only agents that already moved this step are passed in (coma_wrapper.py:97-104)."""
    A = d.num_actions
    for other in moved_positions:
        rel = position_to_index(d, other) - position_to_index(d, position)
        z = collision_rule(A, rel)
        if not z:
            continue
        if A == 6:
            if np.sum(mask) > 1:  # guard evaluated once per matching rule (action_space.py:328-344)
                for a in z:
                    mask[a] = 0
        elif A == 9:
            for a in z:
                mask[a] = 0
                if np.count_nonzero(mask) == 0:
                    mask[a] = 1
        else:
            for a in z:
                mask[a] = 0
    return mask


# --------------------------------------------------------------------------------------
# communication (agent/communication_log.py:12-58)
# --------------------------------------------------------------------------------------


def episode_comm_range(d: Derived, episode: int) -> float:
    if d.fix_range:
        return d.comm_range
    rs = np.random.RandomState(episode)
    return [0, 15, 25, 100][int(rs.randint(4))]


def received_set(positions: Sequence, i: int, comm_range: float, failure_rate: float, draws: Sequence[float]) -> List[int]:
    """Agents whose message agent ``i`` receives (incl. itself).  ``draws[j]`` replaces the one
    np.random.random_sample() the reference consumes per candidate j (communication_log.py:46)."""
    out = []
    pi = np.asarray(positions[i])
    for j in range(len(positions)):
        dist = np.linalg.norm(pi - np.asarray(positions[j]), ord=2)
        ok = dist < 0.001
        if 0.001 <= dist <= comm_range and draws[j] >= failure_rate:
            ok = True
        if ok:
            out.append(j)
    return out


# --------------------------------------------------------------------------------------
# entropy / reward (utils/state.py:14-121, utils/reward.py:11-82)
# --------------------------------------------------------------------------------------


def area_weights(n_src: int, n_dst: int) -> np.ndarray:
    """[n_dst, n_src] weights of the exact area average (definition of INTER_AREA when shrinking)."""
    s = n_src / n_dst
    w = np.zeros((n_dst, n_src))
    for o in range(n_dst):
        lo, hi = o * s, (o + 1) * s
        for i in range(int(math.floor(lo)), min(n_src, int(math.ceil(hi)))):
            w[o, i] = max(0.0, min(hi, i + 1) - max(lo, i)) / s
    return w


def area_resize(src: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize(src, (w, h), interpolation=INTER_AREA) stand-in; keeps the source dtype like OpenCV."""
    w, h = int(dsize[0]), int(dsize[1])
    wr = area_weights(src.shape[0], h)
    wc = area_weights(src.shape[1], w)
    return (wr @ src.astype(np.float64) @ wc.T).astype(src.dtype)


def shannon_entropy(p: np.ndarray) -> np.ndarray:
    """utils/state.py:118-121 -- clips p IN PLACE."""
    p[0.0001 > p] = 0.0001
    p[0.9999 < p] = 0.9999
    return -p * np.log2(p) - (1 - p) * np.log2(1 - p)


def class_weights(target_src: np.ndarray) -> np.ndarray:
    """utils/state.py:60-73 with the hard-coded class_weighting [0, 1]."""
    t = copy.deepcopy(target_src)
    t[t > 0.501] = 1
    t[t < 0.499] = 0
    w = t.copy()
    w[np.round(w, 2) == 0] = 0
    w[np.round(w, 2) == 1] = 1
    w[np.round(w, 2) == 0.5] = 0.5
    return w


def w_entropy_map(d: Derived, map_footprint, local_map, simulated_map, observability: str):
    """get_w_entropy_map + calculate_w_entropy (utils/state.py:14-115).
    Returns (w*H, w, H, w_fp*H_fp or None, grid_map[clipped in place])."""
    if observability not in ("reward", "eval"):
        dsize = (d.space_y, d.space_x)
        grid = area_resize(local_map, dsize)
        if observability == "actor":
            map_footprint = area_resize(map_footprint, dsize)
        simulated_map = area_resize(simulated_map, dsize)
    else:
        grid = local_map.copy()
    weights = class_weights(simulated_map if observability == "eval" else grid)
    se = shannon_entropy(grid)
    w_fp = None
    if observability == "actor":
        wf = class_weights(map_footprint)
        w_fp = wf * shannon_entropy(map_footprint)
    return weights * se, weights, se, w_fp, grid


def utility_reward(d: Derived, before: np.ndarray, after: np.ndarray, simulated_map) -> Tuple[float, float]:
    """utils/reward.py:68-82 -> (absolute, relative)."""
    h_before = w_entropy_map(d, None, before, simulated_map, "reward")[2]
    out = w_entropy_map(d, None, after, simulated_map, "reward")
    reduction = h_before - out[2]
    absolute = np.mean(out[1] * reduction)
    relative = absolute / np.mean(out[1] * h_before)
    return absolute, relative


def global_reward(d: Derived, before, after, simulated_map) -> Tuple[bool, float, float]:
    """utils/reward.py:11-53 -> (done=False, 22*rel-0.5, 10*abs-0.17)."""
    absolute, relative = utility_reward(d, before, after, simulated_map)
    return False, 22 * relative - 0.5, 10 * absolute - 0.17


def reward_sums(d: Derived, before: np.ndarray, after: np.ndarray) -> Tuple[float, float]:
    """(S1, S2) of SURVEY Appendix D in float64: S1 = sum w(a)(H(b)-H(a)), S2 = sum w(a)H(b)."""
    hb = shannon_entropy(before.astype(np.float64).copy())
    a = after.astype(np.float64).copy()
    w = class_weights(a)
    ha = shannon_entropy(a)
    return float(np.sum(w * (hb - ha))), float(np.sum(w * hb))


# --------------------------------------------------------------------------------------
# network inputs (actor/transformations.py:14-176, critic/transformations.py:17-132)
# --------------------------------------------------------------------------------------


def actor_position_map(d: Derived, local_information: Dict, agent_id: int) -> np.ndarray:
    """Egocentric 11x11 plane; centre index 5 is hard-coded by the reference (transformations.py:110-176)."""
    pm = np.ones((d.space_x, d.space_y))
    own = None
    writes = None
    others = []
    for idx in local_information:
        pidx = position_to_index(d, local_information[idx]["position"])
        if idx == agent_id:
            own = pidx
            writes = [[5, 5, (own[2] + 1) / (d.space_z + 1)]]
            if own[0] < 5:
                pm[0:5 - own[0], :] = 0
            if own[1] < 5:
                pm[:, 0:5 - own[1]] = 0
            if own[0] > 5:
                pm[d.space_x - 1 - (own[0] - 6):, :] = 0
            if own[1] > 5:
                pm[:, d.space_y - 1 - (own[1] - 6):] = 0
        else:
            others.append(pidx)
    for o in others:
        writes.append([o[0] - own[0] + 5, o[1] - own[1] + 5, (o[2] + 1) / (d.space_z + 1)])
    for r0, r1, v in writes:
        if 0 <= r0 < d.space_x and 0 <= r1 < d.space_x:
            pm[int(r0), int(r1)] = v
    return pm


def actor_footprint_map(d: Derived, local_information: Dict, agent_id: int) -> np.ndarray:
    """1 in own footprint, 0 in received others' footprints (own wins), 0.5 elsewhere; then resized
    (actor/transformations.py:62-83)."""
    own = local_information[agent_id]["map2communicate"]
    fm = own.copy()
    fm[fm < 0.49] = 1
    fm[fm > 0.51] = 1
    for j in local_information:
        if j == agent_id:
            continue
        m = local_information[j]["map2communicate"]
        fm[m < 0.49] = 0
        fm[m > 0.51] = 0
    fm[own < 0.49] = 1
    fm[own > 0.51] = 1
    return area_resize(fm, (d.space_y, d.space_x))


def actor_observation(d: Derived, local_information: Dict, fused_local_map, simulated_map, agent_id: int, t: int) -> np.ndarray:
    """float64 [11,11,7]: budget, id, position, w-entropy, local w-entropy, prob, footprint
    (actor/transformations.py:14-59)."""
    pm = actor_position_map(d, local_information, agent_id)
    w_ent, _, _, local_w_ent, prob = w_entropy_map(
        d, local_information[agent_id]["footprint_img"], fused_local_map, simulated_map, "actor")
    budget = np.ones_like(pm) * ((d.budget - t) / d.budget)
    aid = np.ones_like(pm) * ((agent_id + 1) / d.n_agents)
    fpm = actor_footprint_map(d, local_information, agent_id)
    return np.dstack([budget, aid, pm, w_ent, local_w_ent, prob, fpm])


def critic_state(d: Derived, global_information: Dict, global_map, actor_obs: np.ndarray, actions: Sequence[int],
                 agent_id: int, simulated_map) -> np.ndarray:
    """float32 [11,11,12] (critic/transformations.py:17-132).  ``actions`` are this step's chosen actions;
    positions/footprints in ``global_information`` are the pre-move ones."""
    pos = np.zeros((d.space_x, d.space_y))
    for j in range(len(global_information)):
        pidx = position_to_index(d, global_information[j]["position"])
        pos[pidx[0], pidx[1]] = (pidx[2] + 1) / d.space_z
    w_ent, _, _, _, prob = w_entropy_map(d, None, global_map, simulated_map, "global")
    act = np.zeros((d.space_x, d.space_y))
    for j in range(d.n_agents):
        if j == agent_id:
            continue
        pidx = position_to_index(d, global_information[j]["position"])
        act[pidx[0], pidx[1]] = (int(actions[j]) + 1) / d.num_actions
    fm = global_information[0]["map2communicate"].copy()
    fm[fm < 0.49] = 1
    fm[fm > 0.51] = 1
    for j in global_information:
        if j == 0:
            continue
        m = global_information[j]["map2communicate"]
        fm[m < 0.49] = 1
        fm[m > 0.51] = 1
    fm = area_resize(fm, (d.space_y, d.space_x))
    return np.dstack((actor_obs, pos[..., None], w_ent, prob, fm[..., None], act[..., None])).astype(np.float32)


# --------------------------------------------------------------------------------------
# one episode with injected randomness (missions/episode_generator.py:38-88, coma_wrapper.py:37-183,
# agent/agent.py:40-104)
# --------------------------------------------------------------------------------------


class OracleEpisode:
    """Steps one environment the way EpisodeGenerator.execute does.

    Randomness is injected:
      correctness(agent_id, s, shape) -> {0,1} array; s = 0 for the start-position sensing, t+1 for step t
      choose_action(agent_id, t, mask, obs) -> int       (replaces actor forward + torch.multinomial)
      comm_draw(i, j, t) -> float in [0,1)               (replaces np.random.random_sample())
    """

    def __init__(self, params: Dict, episode: int, correctness: Callable, choose_action: Callable,
                 comm_draw: Optional[Callable] = None, truth: Optional[np.ndarray] = None,
                 build_features: bool = True, start_positions: Optional[Sequence] = None, exact: bool = False):
        self.d = Derived(params)
        self.d.exact = exact
        self.episode = episode
        self.truth = make_truth(self.d, episode) if truth is None else truth
        self.correctness = correctness
        self.choose_action = choose_action
        self.comm_draw = comm_draw or (lambda i, j, t: 1.0)
        self.build_features = build_features
        self.start_positions = start_positions
        d = self.d
        self.comm_range = episode_comm_range(d, episode)
        self.agents = [dict(local_map=init_prior_map(d), position=None, map2communicate=None,
                            footprint_img=None, rect=None) for _ in range(d.n_agents)]
        self.global_map = self.agents[0]["local_map"].copy()
        self.log: List[Dict] = []

    def _sense(self, i: int, s: int):
        ag = self.agents[i]
        _, fc = project_field_of_view(self.d, ag["position"])
        corr = np.asarray(self.correctness(i, s, tile_shape(fc)))
        lm, _, fc, m2c, fimg = update_grid_map(self.d, self.truth, ag["position"], ag["local_map"], corr)
        ag.update(local_map=lm, rect=fc, map2communicate=m2c, footprint_img=fimg)

    def step(self, t: int) -> Dict:
        d = self.d
        n = d.n_agents
        # ---- build_observations: communicate (t==0: start state + first sensing)
        if t == 0:
            for i in range(n):
                if self.start_positions is not None:
                    self.agents[i]["position"] = np.array(self.start_positions[i])
                else:
                    self.agents[i]["position"] = start_state(d, i, self.episode)
                self._sense(i, 0)
        published = {i: dict(position=self.agents[i]["position"], map2communicate=self.agents[i]["map2communicate"],
                             footprint_img=self.agents[i]["footprint_img"], rect=self.agents[i]["rect"])
                     for i in range(n)}
        positions = [published[i]["position"] for i in range(n)]
        received, observations = [], []
        for i in range(n):
            draws = [self.comm_draw(i, j, t) for j in range(n)]
            ks = received_set(positions, i, self.comm_range, d.failure_rate, draws)
            local_info = {j: published[j] for j in ks}
            self.agents[i]["local_map"] = fuse_map(d, self.agents[i]["local_map"], local_info, i, "local")
            received.append(ks)
            if self.build_features:
                observations.append(actor_observation(d, local_info, self.agents[i]["local_map"], self.truth, i, t))
            else:
                observations.append(None)
        fused_local = [self.agents[i]["local_map"].copy() for i in range(n)]
        # the 11 x 11 area averages that decide the class weights of the feature planes (actor planes 3, 4; critic plane 8): a
        # test that meets a whole-class-weight difference can check that the deciding average sits ON a threshold (0.499 / 0.501)
        decide_local = decide_fp = None
        if self.build_features:
            dsize = (d.space_y, d.space_x)
            decide_local = [area_resize(self.agents[i]["local_map"].astype(np.float64), dsize) for i in range(n)]
            decide_fp = [area_resize(np.asarray(published[i]["footprint_img"], dtype=np.float64), dsize) for i in range(n)]
        # ---- steps: global fusion, sequential act/move/sense, critic input, reward
        critic_map = fuse_map(d, self.global_map, published, None, "global")
        moved, actions, masks = [], [], []
        for i in range(n):
            m = action_mask(d, self.agents[i]["position"])
            m = apply_collision_mask(d, self.agents[i]["position"], m, moved)
            a = int(self.choose_action(i, t, m.copy(), observations[i]))
            self.agents[i]["position"] = action_to_position(d, self.agents[i]["position"], a)
            self._sense(i, t + 1)
            moved.append(self.agents[i]["position"])
            actions.append(a)
            masks.append(m)
        states = None
        if self.build_features:
            states = [critic_state(d, published, critic_map, observations[i], actions, i, self.truth) for i in range(n)]
        _, rel, abs_ = global_reward(d, self.global_map, critic_map, self.truth)
        s1, s2 = reward_sums(d, self.global_map, critic_map)
        self.global_map = critic_map
        rec = dict(t=t, positions=np.array(positions), received=received, observations=observations, states=states,
                   masks=np.array(masks), actions=np.array(actions), next_positions=np.array(moved),
                   rects=np.array([published[i]["rect"] for i in range(n)]),
                   next_rects=np.array([self.agents[i]["rect"] for i in range(n)]),
                   relative_reward=float(rel), absolute_reward=float(abs_), s1=s1, s2=s2,
                   done=(t == d.budget), fused_local=fused_local, global_map=critic_map.copy(),
                   sensed_local=[self.agents[i]["local_map"].copy() for i in range(n)],   # after this step's sensing
                   decide_local=decide_local, decide_fp=decide_fp,
                   decide_global=area_resize(critic_map.astype(np.float64), (d.space_y, d.space_x)) if self.build_features else None)
        self.log.append(rec)
        return rec

    def run(self) -> List[Dict]:
        for t in range(self.d.budget + 1):
            self.step(t)
        return self.log


# --------------------------------------------------------------------------------------
# TD(lambda) targets (batch_memory.py:120-162) -- literal restatement over one agent's transition list
# --------------------------------------------------------------------------------------


def td_lambda_targets(rewards: Sequence[float], dones: Sequence[bool], q_sel: Sequence[float], gamma: float, lam: float):
    """``q_sel[t]`` = target-critic Q(state_t)[action_t].  Returns (td_target[L], discounted_return[L]).

    Lists may span several episodes (the reference clears the memory only after an update); the done flag of
    the *previous* transition gates accumulation, with Python's negative index at t+l-1 == -1 guarded by
    ``t + l == 0``."""
    L = len(rewards)
    td = np.zeros(L)
    dr = np.zeros(L)
    for t in range(L):
        total = 0.0
        disc = 0.0
        for n in range(1, L - t + 1):
            leave = False
            g = 0.0
            disc = 0.0
            for l in range(0, n):
                if (not dones[t + l - 1]) or (t + l == 0):
                    g += gamma ** l * rewards[t + l]
                    disc += gamma ** l * rewards[t + l]
                else:
                    leave = True
                    break
            if leave:
                total += lam ** n * g
                break
            if t + n < L:
                if not (dones[t + n] or (t + n + 1 >= L)):
                    g += gamma ** n * q_sel[t + n]
            total += lam ** (n - 1) * g
        td[t] = (1 - lam) * total
        dr[t] = disc
    return td, dr


# --------------------------------------------------------------------------------------
# COMA counterfactual advantage (actor/learner.py:55-95) on plain arrays
# --------------------------------------------------------------------------------------


def coma_advantage(probs: np.ndarray, q: np.ndarray, mask: np.ndarray, actions: np.ndarray):
    """probs/q/mask [B,A], actions [B] -> (advantage[B], baseline[B], pi_tilde[B,A])."""
    p = probs * mask
    s = p.sum(-1, keepdims=True)
    s = np.where(s < 1e-5, 1e-5, s)
    pn = p / s
    pn = np.where(pn <= 1e-5, 1e-5, pn)
    baseline = (pn * q * mask).sum(-1)
    qa = np.take_along_axis(q, actions[:, None].astype(np.int64), axis=1)[:, 0]
    return qa - baseline, baseline, pn


def epsilon_schedule(params: Dict, num_episode: int) -> float:
    """actor/network.py:53-58"""
    m = params["experiment"]["missions"]
    if num_episode > m["eps_anneal_phase"]:
        return m["eps_min"]
    return m["eps_max"] - num_episode / m["eps_anneal_phase"] * (m["eps_max"] - m["eps_min"])


# --------------------------------------------------------------------------------------
# counter-based RNG shared with the device: Philox4x32-10 (Salmon et al. 2011), vectorised
# --------------------------------------------------------------------------------------

_PH_M0 = np.uint64(0xD2511F53)
_PH_M1 = np.uint64(0xCD9E8D57)
_PH_W0 = 0x9E3779B9
_PH_W1 = 0xBB67AE85
_U32 = np.uint64(0xFFFFFFFF)

DOMAIN_FLIP, DOMAIN_ACTION, DOMAIN_COMM = 0, 1, 2


def philox4x32(c0, c1, c2, c3, k0: int, k1: int):
    """Inputs broadcastable integer arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _U32 for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _PH_M0 * c0
        p1 = _PH_M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & _U32
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & _U32
        c0, c1, c2, c3 = n0 & _U32, n1, n2 & _U32, n3
        k0 = (k0 + _PH_W0) & 0xFFFFFFFF
        k1 = (k1 + _PH_W1) & 0xFFFFFFFF
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def philox_stream_word(agent: int, s: int, domain: int) -> int:
    return (agent & 0xFF) | ((s & 0xFFFF) << 8) | ((domain & 0xFF) << 24)


def philox_flip_threshold(noise: float) -> int:
    """flip iff r < thresh (uint32).  Computed in float64 exactly as the host does."""
    return int(math.floor(noise * 4294967296.0))


def philox_correctness(seed: int, episode: int, agent: int, s: int, rect, gy: int, noise: float) -> np.ndarray:
    """correctness tile for the clipped rect [yu,yd,xl,xr]: cell (x,y) uses counter ((x*gy+y)>>2, episode,
    stream word, 0) and output lane (x*gy+y)&3."""
    yu, yd, xl, xr = rect
    xs = np.arange(xl, xr, dtype=np.int64)[:, None]
    ys = np.arange(yu, yd, dtype=np.int64)[None, :]
    lin = xs * gy + ys
    r = philox4x32(lin >> 2, episode & 0xFFFFFFFF, philox_stream_word(agent, s, DOMAIN_FLIP), (episode >> 32) & 0xFFFFFFFF,
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    lane = lin & 3
    word = np.where(lane == 0, r[0], np.where(lane == 1, r[1], np.where(lane == 2, r[2], r[3])))
    return (word >= np.uint32(philox_flip_threshold(noise))).astype(np.int64)


def philox_action_word(seed: int, episode: int, agent: int, t: int) -> int:
    r = philox4x32(0, episode & 0xFFFFFFFF, philox_stream_word(agent, t, DOMAIN_ACTION), (episode >> 32) & 0xFFFFFFFF,
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return int(r[0])


def philox_comm_draw(seed: int, episode: int, i: int, j: int, t: int) -> float:
    r = philox4x32(j, episode & 0xFFFFFFFF, philox_stream_word(i, t, DOMAIN_COMM), (episode >> 32) & 0xFFFFFFFF,
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return float(int(r[0]) * (1.0 / 4294967296.0))


def uniform_valid_action(word: int, mask: np.ndarray) -> int:
    """Random policy: k-th valid action, k = (word * nvalid) >> 32 (integer arithmetic, bit-exact)."""
    valid = np.flatnonzero(np.asarray(mask) > 0)
    if len(valid) == 0:
        raise ValueError("empty action mask (the reference's torch.multinomial raises here)")
    return int(valid[(int(word) * len(valid)) >> 32])


def sample_masked_action(word: int, probs_masked: np.ndarray) -> int:
    """Inverse-CDF draw over unnormalised float32 weights, sequential float32 accumulation (no FMA)."""
    p = np.asarray(probs_masked, dtype=np.float32)
    total = np.float32(0)
    for v in p:
        total = np.float32(total + v)
    if not total > 0:
        raise ValueError("empty action distribution")
    u = np.float32((int(word) >> 8) * (1.0 / 16777216.0))
    target = np.float32(u * total)
    acc = np.float32(0)
    last = 0
    for a, v in enumerate(p):
        if v > 0:
            last = a
            acc = np.float32(acc + v)
            if acc > target:
                return a
    return last


# --------------------------------------------------------------------------------------
# greedy information-gain baseline (IG_baseline.py:32-325) and its evaluation metrics (utils/utils.py:43-76)
# --------------------------------------------------------------------------------------


def ig_individual(d: Derived, position, mask: np.ndarray, map_state: np.ndarray):
    """IG_baseline.get_individual_ig (:222-289): expected weighted entropy reduction over each candidate footprint.
    Returns (action_positions, information_gains); masked actions get position 0 and gain 0 like the reference."""
    positions, gains = [], []
    for action in range(len(mask)):
        if mask[action] == 0:
            positions.append(0)
            gains.append(0)
            continue
        new_position = action_to_position(d, position, action)
        _, fc = project_field_of_view(d, new_position)
        section = map_state[fc[2]:fc[3], fc[0]:fc[1]].copy()
        noise = noise_of_altitude(new_position[2])
        cw1 = bayes_update(section.copy(), 1 - noise, d.prior)
        cw2 = bayes_update(section.copy(), noise, d.prior)
        cw1[cw1 > 0.501] = 1
        cw1[cw1 < 0.499] = 0
        cw2[cw2 > 0.501] = 1
        cw2[cw2 < 0.499] = 0
        h = shannon_entropy(section)  # clips `section` in place: every later use sees the clipped values
        ig = (section * (h - shannon_entropy(bayes_update(section, 1 - noise, d.prior))) * cw1
              + (1 - section) * (shannon_entropy(section) - shannon_entropy(bayes_update(section, noise, d.prior))) * cw2)
        positions.append(new_position)
        gains.append(np.sum(ig) / 1000)
    return positions, gains


def ig_relative(gain_lists):
    """get_relative_ig (:291-298): per-agent normalisation, in place."""
    # (the reference's gains are NumPy scalars: an agent whose candidates all have zero gain gets 0 / 0 = nan with a warning, not
    #  a ZeroDivisionError; np.argmax then takes the first nan)
    for a in range(len(gain_lists)):
        total = np.float64(sum(gain_lists[a]))
        for k in range(len(gain_lists[a])):
            with np.errstate(divide="ignore", invalid="ignore"):
                gain_lists[a][k] = np.float64(gain_lists[a][k]) / total
    return gain_lists


def ig_cell_utilities(position_lists, rel):
    """get_cell_utilities (:300-322): discount candidate cells other agents also consider; in place, order-dependent."""
    for a in range(len(position_lists)):
        for k1 in range(len(position_lists[a])):
            p1, g1 = position_lists[a][k1], rel[a][k1]
            for b in range(len(position_lists)):
                if b == a:
                    continue
                for k2 in range(len(position_lists[b])):
                    p2, g2 = position_lists[b][k2], rel[b][k2]
                    if np.array_equal(p1, p2) and type(p1) is np.ndarray:
                        rel[a][k1] = g1 * (1 - g2)
    return rel


def target_entropy(d: Derived, gmap: np.ndarray, truth: np.ndarray) -> float:
    """Mean entropy over the target cells ("eval" weights come from the ground truth; IG_baseline.py:84-97)."""
    wh = w_entropy_map(d, None, gmap, truth, "eval")[0]
    masked = wh.copy()
    masked[truth == 0] = 0
    counts = np.unique(truth, return_counts=True)[1]
    return float(np.sum(masked) / counts[-1])


# Cells whose observations cancel exactly sit at p = 0.5 +- rounding noise (|log-odds| < 1e-7), and which side of the threshold
# they fall on is decided by the last bit of whoever computed them.  Every other cell is DECIDABLE: at least one measurement's
# log-odds (>= 0.5) away from 0.  The counts under the two thresholds +-1e-5 in log-odds separate the two kinds, and are integers
# that any correct implementation must reproduce exactly (the device's ippm_f1_counts takes the same threshold).
F1_DECIDABLE_LOGODDS = 1e-5
_F1_COUNTS_LOG = None


def f1_counts(gmap: np.ndarray, truth: np.ndarray, logodds_thr: float = 0.0):
    """(tp, fp, fn) of class 1 for the map thresholded at log-odds > logodds_thr (0 <=> p > 0.5: utils/utils.py:64-76)."""
    p = np.asarray(gmap, dtype=np.float64)
    with np.errstate(divide="ignore"):
        pred = (np.log(p) - np.log1p(-p)) > logodds_thr
    t = np.asarray(truth) == 1
    return int(np.sum(pred & t)), int(np.sum(pred & ~t)), int(np.sum(~pred & t))


class record_f1_counts:
    """``with record_f1_counts() as log:`` -- every f1_target() evaluated inside appends ((tp, fp, fn) at +1e-5, (tp, fp, fn) at -1e-5)."""

    def __enter__(self):
        global _F1_COUNTS_LOG
        self._saved, _F1_COUNTS_LOG = _F1_COUNTS_LOG, []
        return _F1_COUNTS_LOG

    def __exit__(self, *exc):
        global _F1_COUNTS_LOG
        _F1_COUNTS_LOG = self._saved


def f1_target(gmap: np.ndarray, truth: np.ndarray) -> float:
    """F1 of class 1 of the map thresholded at 0.5 (utils/utils.py:64-76, sklearn f1_score(average=None)[1])."""
    if _F1_COUNTS_LOG is not None:
        _F1_COUNTS_LOG.append((f1_counts(gmap, truth, F1_DECIDABLE_LOGODDS), f1_counts(gmap, truth, -F1_DECIDABLE_LOGODDS)))
    pred = gmap > 0.5
    t = truth == 1
    tp, fp, fn = np.sum(pred & t), np.sum(pred & ~t), np.sum(~pred & t)
    den = 2 * tp + fp + fn
    return float(2 * tp / den) if den > 0 else 0.0


class OracleIGBaseline:
    """IG_baseline.execute (:56-220) with injected sensing randomness (see OracleEpisode for the conventions)."""

    def __init__(self, params: Dict, episode: int, correctness: Callable, comm_draw: Optional[Callable] = None, exact: bool = False):
        self.d = Derived(params)
        self.d.exact = exact
        self.episode = episode
        self.truth = make_truth(self.d, episode)
        self.correctness = correctness
        self.comm_draw = comm_draw or (lambda i, j, t: 1.0)
        self.communication = params["experiment"]["baselines"]["information_gain"]["communication"]
        self.comm_range = episode_comm_range(self.d, episode)

    def execute(self):
        d, n = self.d, self.d.n_agents
        agents = [dict(local_map=init_prior_map(d), position=None, map2communicate=None, stage=0) for _ in range(n)]

        def sense(i):
            ag = agents[i]
            _, fc = project_field_of_view(d, ag["position"])
            corr = np.asarray(self.correctness(i, ag["stage"], tile_shape(fc)))
            ag["stage"] += 1
            lm, _, _, m2c, _ = update_grid_map(d, self.truth, ag["position"], ag["local_map"], corr)
            ag.update(local_map=lm, map2communicate=m2c)

        current_global = agents[0]["local_map"].copy()
        entropies = [target_entropy(d, current_global, self.truth)]
        f1s = [f1_target(current_global, self.truth)]
        rel_rewards, abs_rewards, altitudes, positions_log, gains_log, actions_log = [], [], [], [], [], []
        for t in range(d.budget + 1):
            if t == 0:
                for i in range(n):
                    agents[i]["position"] = start_state(d, i, self.episode)
                    sense(i)
            published = {i: dict(position=agents[i]["position"], map2communicate=agents[i]["map2communicate"]) for i in range(n)}
            pos = [published[i]["position"] for i in range(n)]
            for i in range(n):
                ks = received_set(pos, i, self.comm_range, d.failure_rate, [self.comm_draw(i, j, t) for j in range(n)])
                agents[i]["local_map"] = fuse_map(d, agents[i]["local_map"], {j: published[j] for j in ks}, i, "local")
            if t == 0:
                positions_log.append([p.copy() for p in pos])
                current_global = fuse_map(d, current_global, published, None, "global")
            next_positions, pos_lists, gain_lists = [], [], []
            for i in range(n):
                m = action_mask(d, agents[i]["position"])
                m = apply_collision_mask(d, agents[i]["position"], m, next_positions)
                ap, ig = ig_individual(d, agents[i]["position"], m, agents[i]["local_map"])
                pos_lists.append(ap)
                gain_lists.append(ig)
                next_positions.append(agents[i]["position"])
            gains_log.append([list(g) for g in gain_lists])
            rel = ig_relative(gain_lists)
            util = ig_cell_utilities(pos_lists, rel) if self.communication else rel
            step_actions, alts, m2cs = [], [], []
            for i in range(n):
                a = int(np.argmax(util[i]))
                agents[i]["position"] = action_to_position(d, agents[i]["position"], a)
                sense(i)
                m2cs.append(agents[i]["map2communicate"])
                next_positions.append(agents[i]["position"])
                alts.append(int(agents[i]["position"][2]))
                step_actions.append(a)
            actions_log.append(step_actions)
            positions_log.append([p.copy() for p in next_positions])
            altitudes.append(alts)
            next_global = fuse_map(d, current_global, m2cs, None, "global")
            current_global = next_global.copy()
            _, rel_r, abs_r = global_reward(d, current_global, next_global, self.truth)  # constants
            rel_rewards.append(rel_r)
            abs_rewards.append(abs_r)
            entropies.append(target_entropy(d, next_global, self.truth))
            f1s.append(f1_target(next_global, self.truth))
        self.agents, self.global_map = agents, current_global
        return dict(relative_return=sum(rel_rewards), absolute_return=sum(abs_rewards), altitudes=altitudes, entropies=entropies,
                    f1=f1s, gains=gains_log, actions=actions_log, positions=positions_log)


# --------------------------------------------------------------------------------------
# comparison / deployment scripts (random_baseline.py, lawn_mower.py, coma_test.py): metric curves
# --------------------------------------------------------------------------------------


def lawnmower_paths(altitude: int) -> List[np.ndarray]:
    """positions1..8 of lawn_mower.py:46-203: two row-wise and two column-wise 15-cell sweeps, each listed twice."""
    run = list(range(10, 45, 5))

    def lane(c, along_x):
        cells = [(v, c) for v in run] + [(40, c + 5)] + [(v, c + 10) for v in run[::-1]]
        return np.array([[a, b, altitude] if along_x else [b, a, altitude] for a, b in cells])

    four = [lane(10, True), lane(30, True), lane(10, False), lane(30, False)]
    return four + [p.copy() for p in four]


def shared_map_curves(d: Derived, truth: np.ndarray, visits, correctness: Callable):
    """random_baseline.py:40-124 / lawn_mower.py:231-313: ONE map, updated in place by every platform's sensing
    (Mapping.update_grid_map on the shared map), metrics after each step.  ``visits[s]`` = positions sensed in step s (in
    order); ``correctness(k)`` = draws of the k-th sensing overall.  -> (entropies, f1s), initial prior entry first."""
    m = init_prior_map(d)
    ent, f1 = [target_entropy(d, m.copy(), truth)], [f1_target(m, truth)]
    k = 0
    for step in visits:
        for pos in step:
            _, fc = project_field_of_view(d, pos)
            corr = np.asarray(correctness(k)).reshape(fc[3] - fc[2], fc[1] - fc[0])
            m = update_grid_map(d, truth, pos, m.copy(), corr)[0]
            k += 1
        ent.append(target_entropy(d, m.copy(), truth))
        f1.append(f1_target(m, truth))
    return ent, f1


def deployment_curves(params: Dict, episode: int, actions: Callable, correctness: Callable, comm_draw: Optional[Callable] = None):
    """coma_test.py:98-196 with the greedy choices given (``actions(t, i)``): the episode runs as in training (publish,
    receive, fuse local maps, move, sense) but the global map fuses the start measurements at t = 0 and then, every
    step, the measurements taken right after the move (no one-step lag), and both metrics are logged per step.
    The measurements go in as the script passes them: through the dict branch of fuse_map at t = 0 (float32 cast) and
    as a bare list afterwards (no cast: a measurement published from an already-fused, hence promoted, local map stays
    float64 under NumPy 2, and so do that step's updates -- this decides the class of exactly-cancelled cells).
    ``correctness(s, i)``: draws of agent i's sensing stage s.  -> (entropies, f1s, positions[T+2? no: T+1 stages])"""
    seen = {}
    ep = OracleEpisode(params, episode, lambda i, s, shape: np.asarray(correctness(s, i)).reshape(shape),
                       lambda i, t, m, o: actions(t, i), comm_draw=comm_draw, build_features=False)
    real_sense = ep._sense

    def sense(i, s):
        real_sense(i, s)
        seen[(i, s)] = ep.agents[i]["map2communicate"]

    ep._sense = sense
    d, n = ep.d, ep.d.n_agents
    g = init_prior_map(d)
    ent, f1 = [target_entropy(d, g.copy(), ep.truth)], [f1_target(g, ep.truth)]
    stages = []
    for t in range(d.budget + 1):
        rec = ep.step(t)
        if t == 0:
            stages.append(rec["positions"])
            g = fuse_map(d, g, {i: dict(map2communicate=seen[(i, 0)]) for i in range(n)}, None, "global")
        stages.append(rec["next_positions"])
        g = fuse_map(d, g, [seen[(i, t + 1)] for i in range(n)], None, "global")
        ent.append(target_entropy(d, g.copy(), ep.truth))
        f1.append(f1_target(g, ep.truth))
    return ent, f1, np.array(stages)
