"""Golden-vector generator: imports the *reference* (read-only, /root/reference) in the build container,
runs its own functions on seeded inputs and writes small .npz fixtures to tests/golden/.

BUILD-CONTAINER ONLY.  /root/reference does not exist on the GPU box; nothing at test/bench time runs
this script.  Fixtures are data (inputs + the reference's outputs), never reference source.

Shims injected before import: ``cv2`` (only resize/INTER_AREA are used; replaced by the exact
area average of oracle.ipp_oracle.area_resize -> parity unpinned versus real OpenCV), a no-op
``torch.utils.tensorboard.SummaryWriter`` and an empty ``seaborn``.

    python oracle/make_golden.py            # regenerates every fixture
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import ipp_oracle as O  # noqa: E402
from configs import make_params, synthetic_minibatch  # noqa: E402


def _install_shims():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_AREA = 3
    cv2.resize = lambda src, dsize, interpolation=None: O.area_resize(np.asarray(src), dsize)
    sys.modules["cv2"] = cv2
    import torch.utils  # noqa: F401

    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:  # noqa: D401
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    sys.modules["seaborn"] = types.ModuleType("seaborn")
    sys.path[:0] = [REF, os.path.join(REF, "marl_framework")]


_install_shims()
import torch  # noqa: E402
import yaml  # noqa: E402

from marl_framework.agent.action_space import AgentActionSpace  # noqa: E402
from marl_framework.agent.communication_log import CommunicationLog  # noqa: E402
from marl_framework.agent.state_space import AgentStateSpace  # noqa: E402
from marl_framework.batch_memory import BatchMemory  # noqa: E402
from marl_framework.coma_wrapper import COMAWrapper  # noqa: E402
from marl_framework.mapping import ground_truths  # noqa: E402
from marl_framework.mapping.grid_maps import GridMap  # noqa: E402
from marl_framework.mapping.mappings import Mapping  # noqa: E402
from marl_framework.mapping.simulations import Simulation  # noqa: E402
from marl_framework.missions.episode_generator import EpisodeGenerator  # noqa: E402
from marl_framework.sensors import Sensor  # noqa: E402
from marl_framework.sensors.cameras import Camera  # noqa: E402
from marl_framework.sensors.models import SensorModel  # noqa: E402
from marl_framework.sensors.models.sensor_models import AltitudeSensorModel  # noqa: E402
from marl_framework.utils import reward as ref_reward  # noqa: E402
from marl_framework.utils import state as ref_state  # noqa: E402
from marl_framework.utils.utils import TransitionCOMA, get_fixed_footprint_coordinates  # noqa: E402

PARAM_SETS = ["default", "small", "c2", "c4", "c5"]


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def check_schema():
    """The package's default params.yaml must agree with the reference's on every key."""
    with open(os.path.join(REF, "marl_framework", "params.yaml"), "rb") as f:
        ref = yaml.load(f.read(), Loader=yaml.Loader)
    mine = make_params("default")
    mine["environment"]["num_envs"] = ref["environment"]["num_envs"]
    mine["experiment"]["title"] = ref["experiment"]["title"]
    assert mine == ref, "params.yaml schema/defaults drifted from the reference"


# ------------------------------------------------------------------ 1-4: constants, starts, truth, footprints
def gen_derived_and_footprints():
    out = {}
    for name in PARAM_SETS:
        p = make_params(name)
        gm = GridMap(p)
        ss = AgentStateSpace(p)
        cam = Camera(p, AltitudeSensorModel(p), gm)
        out[f"{name}_res"] = np.array([gm.resolution_x, gm.resolution_y])
        out[f"{name}_dims"] = np.array([gm.x_dim, gm.y_dim])
        out[f"{name}_space"] = np.array(ss.space_dim)
        full, clip, fixed = [], [], []
        for x in range(ss.space_x_dim):
            for y in range(ss.space_y_dim):
                for z in range(ss.space_z_dim):
                    pos = ss.index_to_position([x, y, z])
                    f, c = cam.project_field_of_view(pos, gm.resolution_x, gm.resolution_y)
                    full.append(f)
                    clip.append(c)
                    fixed.append(get_fixed_footprint_coordinates(f, c))
        out[f"{name}_fp_full"] = np.array(full, dtype=np.int32)
        out[f"{name}_fp_clip"] = np.array(clip, dtype=np.int32)
        out[f"{name}_fp_fixed"] = np.array(fixed, dtype=np.int32)
    save("derived_footprints", **out)


def gen_start_states():
    p = make_params("default")
    ss = AgentStateSpace(p)
    st = np.array([[ss.get_random_agent_state(a, e) for a in range(16)] for e in range(1, 65)], dtype=np.int32)
    p7 = make_params("default", environment__seed=7)
    ss7 = AgentStateSpace(p7)
    st7 = np.array([[ss7.get_random_agent_state(a, e) for a in range(4)] for e in range(1, 17)], dtype=np.int32)
    save("start_states", seed3=st, seed7=st7)


def gen_truth():
    sp = []
    for e in range(1, 4097):
        np.random.seed(e)
        s = np.random.randint(4)
        pc = np.random.randint(30, 61)
        sp.append((s, pc))
    fields = {}
    for e, (r, c) in [(1, (64, 64)), (2, (64, 48)), (3, (33, 64)), (5, (40, 40))]:
        # signature is (pk, x_dim, y_dim, episode) -> array (y_dim, x_dim)
        fields[f"field_e{e}_{r}x{c}"] = ground_truths.gaussian_random_field(lambda k: k ** (-5.0), c, r, e).astype(np.uint8)
    save("truth", split_pct=np.array(sp, dtype=np.int32), **fields)


def gen_terrain():
    """The random field the reference synthesises for every episode and then overwrites with the half-plane split
    (ground_truths.py:25-40).  It is a local of gaussian_random_field, so the inverse-FFT output is observed through
    a spy on np.fft.ifft2 while the reference runs; normalise + threshold (its lines 32-40) are then applied here."""
    out = {}
    real = np.fft.ifft2
    for e, (r, c) in [(3, (64, 64)), (7, (45, 45)), (11, (48, 80))]:
        seen = []

        def spy(a, *args, _seen=seen, **kw):
            res = real(a, *args, **kw)
            _seen.append(res)
            return res

        np.fft.ifft2 = spy
        try:
            ground_truths.gaussian_random_field(lambda k: k ** (-5.0), c, r, e)
        finally:
            np.fft.ifft2 = real
        f = seen[-1].real
        f = (f - np.min(f)) / (np.max(f) - np.min(f))
        out[f"field_e{e}_{r}x{c}"] = np.packbits((f >= 0.5).astype(np.uint8), axis=None)
    save("terrain", **out)


# ------------------------------------------------------------------ 5: masks
def gen_masks():
    out = {}
    rng = np.random.RandomState(11)
    for A in (4, 6, 9, 27):
        over = dict(experiment__constraints__num_actions=A)
        if A in (4, 9):  # 2-D variants are only self-consistent with a single altitude level
            over.update(experiment__constraints__min_altitude=15, experiment__constraints__max_altitude=15)
        p = make_params("default", **over)
        asp, ss = AgentActionSpace(p), AgentStateSpace(p)
        masks, poss = [], []
        for x in range(ss.space_x_dim):
            for y in range(ss.space_y_dim):
                for z in range(ss.space_z_dim):
                    pos = ss.index_to_position([x, y, z])
                    m, _ = asp.get_action_mask(pos)
                    masks.append(np.asarray(m, dtype=np.float64))
                    poss.append(pos)
        out[f"a{A}_pos"] = np.array(poss, dtype=np.int32)
        out[f"a{A}_mask"] = np.array(masks)
        out[f"a{A}_moves"] = np.array([asp.action_to_position(np.array([25, 25, p["experiment"]["constraints"]["min_altitude"] +
                                                                          (5 if ss.space_z_dim > 1 else 0)]), a) for a in range(A)], dtype=np.int32)
        # collision cases: own position, 1-3 already-moved others near it
        cp, co, cn, cm_in, cm_out = [], [], [], [], []
        for _ in range(1500):
            ix = rng.randint(0, 11, size=2)
            iz = rng.randint(0, ss.space_z_dim)
            pos = ss.index_to_position([ix[0], ix[1], iz])
            n_o = rng.randint(1, 4)
            others = np.zeros((3, 3), dtype=np.int64)
            for k in range(n_o):
                d = rng.randint(-2, 3, size=2)
                oz = rng.randint(0, ss.space_z_dim)
                oi = np.clip(ix + d, 0, 10)
                others[k] = ss.index_to_position([oi[0], oi[1], oz])
            m, _ = asp.get_action_mask(pos)
            m = np.asarray(m, dtype=np.float64).copy()
            cm_in.append(m.copy())
            m2 = asp.apply_collision_mask(pos, m, [others[k] for k in range(n_o)], ss)
            cp.append(pos)
            co.append(others)
            cn.append(n_o)
            cm_out.append(np.asarray(m2, dtype=np.float64))
        out[f"a{A}_col_pos"] = np.array(cp, dtype=np.int32)
        out[f"a{A}_col_others"] = np.array(co, dtype=np.int32)
        out[f"a{A}_col_n"] = np.array(cn, dtype=np.int32)
        out[f"a{A}_col_in"] = np.array(cm_in)
        out[f"a{A}_col_out"] = np.array(cm_out)
    save("masks", **out)


# ------------------------------------------------------------------ 6: comm
def gen_comm():
    rng = np.random.RandomState(5)
    out = {}
    cases = []
    for rg, fail in [(0, 0.0), (15, 0.0), (25, 0.0), (100, 0.0), (25, 0.4)]:
        for _ in range(40):
            p = make_params("default", experiment__uav__communication_range=rg, experiment__uav__failure_rate=fail)
            n = 4
            pos = np.stack([rng.randint(0, 11, size=(n,)) * 5, rng.randint(0, 11, size=(n,)) * 5, rng.randint(1, 4, size=(n,)) * 5], 1)
            log = CommunicationLog(p, 1)
            for a in range(n):
                log.store_agent_message({"position": pos[a]}, a)
            draws = rng.random_sample((n, n))
            it = iter(draws.flatten())
            real = np.random.random_sample
            np.random.random_sample = lambda: next(it)
            try:
                rec = np.zeros((n, n), dtype=np.uint8)
                for a in range(n):
                    for j in log.get_messages(a).keys():
                        rec[a, j] = 1
            finally:
                np.random.random_sample = real
            cases.append((rg, fail, pos, draws, rec))
    out["range"] = np.array([c[0] for c in cases], dtype=np.float64)
    out["failure"] = np.array([c[1] for c in cases])
    out["pos"] = np.array([c[2] for c in cases], dtype=np.int32)
    out["draws"] = np.array([c[3] for c in cases])
    out["received"] = np.array([c[4] for c in cases])
    # per-episode range when fix_range is False
    p = make_params("default", experiment__uav__fix_range=False)
    out["episode_range"] = np.array([CommunicationLog(p, e).communication_range for e in range(1, 65)], dtype=np.float64)
    save("comm", **out)


# ------------------------------------------------------------------ 7/8: bayes + measurement
def _mapping(p, episode=1):
    gm = GridMap(p)
    return Mapping(gm, Sensor(SensorModel(), gm), p, episode)


def gen_bayes_measurement():
    rng = np.random.RandomState(3)
    out = {}
    for prior in (0.5, 0.3):
        p = make_params("small", mapping__prior=prior)
        mp = _mapping(p)
        x = rng.random_sample((64, 64)).astype(np.float32)
        x[0, :8] = [0.0, 1.0, 0.99995, 0.00005, 0.9999, 0.0001, 0.5, 0.999999]
        ys = np.float32(np.round(rng.choice([0.01, 0.99, 0.265, 0.735, 0.375, 0.625], size=(64, 64)), 3))
        out[f"p{prior}_x"] = x.copy()
        out[f"p{prior}_y"] = ys
        out[f"p{prior}_out"] = mp.apply_update(x.copy(), ys, "train")
        # chained saturation: 20 updates of the same cell set
        chain = np.full((6, 4), prior, dtype=np.float32)
        yv = np.float32(np.round(np.array([0.01, 0.99, 0.265, 0.735, 0.375, 0.625]), 3))[:, None] * np.ones((1, 4), dtype=np.float32)
        hist = []
        for _ in range(20):
            chain = np.float32(mp.apply_update(chain, yv, "train"))
            hist.append(chain.copy())
        out[f"p{prior}_chain"] = np.array(hist)
    # measurement with injected correctness
    p = make_params("small")
    sim = Simulation(p, None, 1, AltitudeSensorModel(p))
    truth = (rng.random_sample((40, 50)) > 0.5).astype(np.float64)
    real = torch.multinomial
    for alt in (5, 10, 15):
        corr = (rng.random_sample((40, 50)) > 0.3).astype(np.int64)
        torch.multinomial = lambda w, n, replacement=True, _c=corr: torch.from_numpy(_c.flatten())
        try:
            sim.simulated_map = truth
            meas = sim.get_measurement(alt, [0, 50, 0, 40], "train")
        finally:
            torch.multinomial = real
        out[f"meas_corr_{alt}"] = corr.astype(np.uint8)
        out[f"meas_out_{alt}"] = meas
    out["meas_truth"] = truth.astype(np.uint8)
    save("bayes_measurement", **out)


# ------------------------------------------------------------------ 9: entropy + reward
def gen_entropy_reward():
    rng = np.random.RandomState(9)
    p = make_params("small")
    ss = AgentStateSpace(p)
    out = {}
    shp = (66, 55)  # the functions are shape-agnostic; non-square, non-integer resize scale
    for k in range(3):
        before = rng.random_sample(shp).astype(np.float32)
        before[rng.random_sample(shp) < 0.3] = 0.5
        after = before.copy()
        sel = rng.random_sample(shp) < 0.4
        after[sel] = rng.random_sample(sel.sum()).astype(np.float32)
        after[:2, :4] = [[0.0, 1.0, 0.99999, 0.5], [0.4995, 0.5005, 0.502, 0.498]]
        truth = (rng.random_sample(shp) > 0.5).astype(np.float64)
        out[f"before{k}"] = before
        out[f"after{k}"] = after
        out[f"truth{k}"] = truth.astype(np.uint8)
        for mode in ("reward", "eval", "global"):
            r = ref_state.get_w_entropy_map(None, after.copy(), truth, mode, ss)
            out[f"{mode}{k}_wH"] = r[0]
            out[f"{mode}{k}_w"] = r[1]
            out[f"{mode}{k}_H"] = r[2]
            out[f"{mode}{k}_p"] = r[4]
        done, rel, ab = ref_reward.get_global_reward(before.copy(), after.copy(), "COMA", None, truth, ss, None, None, 0, 14)
        out[f"reward{k}"] = np.array([rel, ab], dtype=np.float64)
    save("entropy_reward", **out)


# ------------------------------------------------------------------ 10/11: recorded episode
class Recorder:
    """Wraps torch.multinomial / np.random.random_sample to record every draw the reference makes."""

    def __init__(self):
        self.correctness = []
        self.actions = []
        self.comm = []
        self._mn = torch.multinomial
        self._rs = np.random.random_sample

    def __enter__(self):
        def mn(w, n, replacement=False, **k):
            r = self._mn(w, n, replacement=replacement, **k)
            if w.numel() == 2 and n > 1:
                self.correctness.append(r.numpy().astype(np.uint8).copy())
            else:
                self.actions.append(int(r.item()))
            return r

        def rs(*a, **k):
            v = self._rs(*a, **k)
            self.comm.append(float(v))
            return v

        torch.multinomial = mn
        np.random.random_sample = rs
        return self

    def __exit__(self, *exc):
        torch.multinomial = self._mn
        np.random.random_sample = self._rs


def run_reference_episode(p, episode, tag, mode="train", snapshots=True, local_agents=None):
    torch.manual_seed(1234 + episode)
    np.random.seed(4321 + episode)
    wrapper = COMAWrapper(p, None)
    bm = BatchMemory(p, wrapper)
    gm = GridMap(p)
    eg = EpisodeGenerator(p, None, gm, Sensor(SensorModel(), gm))
    stash = {}
    real_init = eg.init_agents

    def init_agents(mapping, cw):
        ag = real_init(mapping, cw)
        stash["agents"] = ag
        return ag

    eg.init_agents = init_agents
    gmaps = []
    real_steps = wrapper.steps

    def steps(*a, **k):
        r = real_steps(*a, **k)
        gmaps.append(np.asarray(r[8], dtype=np.float32).copy())
        return r

    wrapper.steps = steps
    with Recorder() as rec:
        ret = eg.execute(episode, bm, wrapper, mode)
    n = p["experiment"]["missions"]["n_agents"]
    T = p["experiment"]["constraints"]["budget"] + 1
    tr = bm.transitions
    obs = np.array([[tr[a][t].observation.numpy() for a in range(n)] for t in range(T)])
    st = np.array([[tr[a][t].state.numpy() for a in range(n)] for t in range(T)])
    act = np.array([[int(tr[a][t].action) for a in range(n)] for t in range(T)], dtype=np.int32)
    msk = np.array([[np.asarray(tr[a][t].mask.cpu().numpy(), dtype=np.float64) for a in range(n)] for t in range(T)])
    rew = np.array([[float(tr[a][t].reward) for a in range(n)] for t in range(T)])
    done = np.array([[bool(tr[a][t].done) for a in range(n)] for t in range(T)])
    pos = np.array(ret[5], dtype=np.int32)  # [T+1, n, 3]
    assert pos.shape == (T + 1, n, 3), pos.shape
    # correctness draws: n at reset, then n per step, in agent order
    assert len(rec.correctness) == n * (T + 1), len(rec.correctness)
    packed = [np.packbits(c) for c in rec.correctness]
    lens = np.array([len(c) for c in rec.correctness], dtype=np.int32)
    agents = stash["agents"]
    arrays = dict(
        episode=np.array(episode), obs=obs, state=st, actions=act, masks=msk, rewards=rew, done=done, positions=pos,
        episode_return=np.array(ret[0]), abs_return=np.array(ret[2]), episode_rewards=np.array(ret[1]),
        truth=np.asarray(ret[3]).astype(np.uint8),
        corr_packed=np.concatenate(packed), corr_lens=lens,
        comm_draws=np.array(rec.comm), sampled_actions=np.array(rec.actions, dtype=np.int32),
        final_local=np.array([np.asarray(a.local_map, dtype=np.float32) for a in agents]),
        final_global=gmaps[-1],
        eps=np.array(ret[7]),
    )
    if snapshots:   # (left out at 493 x 493 to keep the fixture small)
        arrays.update(global_t0=gmaps[0], global_t7=gmaps[7])
    if local_agents is not None:   # (8 UAVs x 512 x 512: the final local maps of a few agents only)
        arrays["final_local"] = arrays["final_local"][list(local_agents)]
        arrays["final_local_agents"] = np.array(list(local_agents), dtype=np.int32)
    save(tag, **arrays)


def gen_episodes():
    run_reference_episode(make_params("c2"), 1, "episode_c2_e1")
    run_reference_episode(make_params("small", experiment__missions__n_agents=3, experiment__uav__fix_range=False,
                                      experiment__uav__failure_rate=0.3, experiment__constraints__num_actions=27), 6,
                          "episode_small27_e6")
    run_reference_episode(make_params("small", experiment__missions__n_agents=5, experiment__uav__communication_range=15), 3,
                          "episode_small5_e3", mode="eval")


def gen_episode_default_grid():
    """One episode at the reference's own default grid (493 x 493: the only BASELINE-relevant grid whose 11 feature bins are
    not whole cells wide), 2 UAVs as BASELINE config 1: pins the assembled actor / critic planes there."""
    run_reference_episode(make_params("c1"), 2, "episode_default_e2", snapshots=False)


def gen_episode_prior_and_c4():
    """Two more whole episodes of the reference itself: mapping.prior = 0.3 (every fused message shifts every cell of the grid:
    the explicit slow path of the fusion) and BASELINE config 4's team and grid (8 UAVs, 512 x 512: plans of up to nine ops)."""
    run_reference_episode(make_params("small", mapping__prior=0.3, experiment__missions__n_agents=3), 4, "episode_small_prior03_e4",
                          snapshots=False)
    run_reference_episode(make_params("c4"), 2, "episode_c4_e2", snapshots=False, local_agents=(0, 5))


# ------------------------------------------------------------------ 12: TD(lambda)
class _TableCritic:
    """Stands in for the target critic: 'state' is a 1-element tensor holding an index into a Q table."""

    def __init__(self, table):
        self.table = table

    def forward(self, state):
        return torch.tensor(self.table[int(state.item())]), None


def gen_td_lambda():
    rng = np.random.RandomState(21)
    p = make_params("default")
    n, T, n_ep = 4, 15, 5
    L = T * n_ep
    table = rng.standard_normal((n * L, 6)).astype(np.float32)
    bm = BatchMemory(p, None)
    rewards = rng.standard_normal((n, L))
    actions = rng.randint(0, 6, size=(n, L))
    for a in range(n):
        for i in range(L):
            bm.add(a, state=torch.tensor([a * L + i]), action=torch.tensor([actions[a, i]]), reward=float(rewards[a, i]),
                   done=(i % T == T - 1))
    bm.build_td_targets(_TableCritic(table))
    td = np.array([[float(bm.transitions[a][i].td_target) for i in range(L)] for a in range(n)])
    dr = np.array([[float(bm.transitions[a][i].discounted_return) for i in range(L)] for a in range(n)])
    qsel = np.array([[table[a * L + i, actions[a, i]] for i in range(L)] for a in range(n)])
    dones = np.array([[i % T == T - 1 for i in range(L)] for a in range(n)])
    # a single-episode list as well (L = 15)
    bm1 = BatchMemory(p, None)
    for i in range(T):
        bm1.add(0, state=torch.tensor([i]), action=torch.tensor([actions[0, i]]), reward=float(rewards[0, i]), done=(i == T - 1))
    bm1.build_td_targets(_TableCritic(table))
    td1 = np.array([float(bm1.transitions[0][i].td_target) for i in range(T)])
    dr1 = np.array([float(bm1.transitions[0][i].discounted_return) for i in range(T)])
    save("td_lambda", rewards=rewards, dones=dones, qsel=qsel, td=td, dr=dr, td_single=td1, dr_single=dr1,
         gamma=np.array(p["networks"]["gamma"]), lam=np.array(p["networks"]["lambda"]))


# ------------------------------------------------------------------ 13: one COMA minibatch through the reference learners
def gen_coma_step():
    p = make_params("c2")
    torch.manual_seed(77)
    wrapper = COMAWrapper(p, None)
    B = p["networks"]["batch_size"]
    obs, state, actions, masks, td = synthetic_minibatch(B, 6, seed=8)
    batch = [TransitionCOMA(torch.tensor(state[i]), torch.tensor(obs[i]), torch.tensor([actions[i]]), torch.tensor(masks[i]),
                            0.0, False, torch.tensor([td[i]]), torch.tensor([0.0])) for i in range(B)]
    with torch.no_grad():
        q0, _ = wrapper.critic_network.forward(torch.tensor(state))
        pi0, _ = wrapper.actor_network.forward(torch.tensor(obs).float(), 0.3)
    q_values, cm = wrapper.critic_learner.learn(0, [batch], 0)
    _, am = wrapper.actor_learner.learn([batch], q_values, 0.3)
    with torch.no_grad():
        pi1, _ = wrapper.actor_network.forward(torch.tensor(obs).float(), 0.3)
    save("coma_step", mb_seed=np.array(8), net_seed=np.array(77), eps=np.array(0.3),
         q0=q0.numpy(), pi0=pi0.numpy(), q_new=q_values[0].numpy(), critic_loss=np.array(float(cm[0])),
         actor_loss=np.array(float(am[0])), adv_mean=np.array(float(am[1])), adv_std=np.array(float(am[2])), pi1=pi1.numpy(),
         critic_fc3_b=wrapper.critic_network.fc3.bias.detach().numpy(), actor_fc3_b=wrapper.actor_network.fc3.bias.detach().numpy(),
         critic_metrics=np.array([float(v) for v in cm[:13]] + [float(v) for v in cm[13]]),   # coma_mission.py:270-345 order
         actor_metrics=np.array([float(v) for v in am[:7]] + [float(v) for v in am[7]]),
         n_actor_params=np.array(sum(x.numel() for x in wrapper.actor_network.parameters())),
         n_critic_params=np.array(sum(x.numel() for x in wrapper.critic_network.parameters())))


# ------------------------------------------------------------------ IG baseline (BASELINE config 1)
def gen_ig_baseline():
    import marl_framework.IG_baseline as ref_ig
    from marl_framework import constants
    for tag, params, episode in (("ig_c1_e1", make_params("c1"), 1),
                                 ("ig_small3_e4", make_params("small", experiment__missions__n_agents=3), 4)):
        torch.manual_seed(99 + episode)
        np.random.seed(77 + episode)
        with Recorder() as rec:
            ig = ref_ig.IG_baseline(params, None, episode)
            gains = []
            real = ig.get_individual_ig

            def spy(position, mask, map_state, _real=real, _g=gains):
                ap, g = _real(position, mask, map_state)
                _g.append([float(v) for v in g])
                return ap, g

            ig.get_individual_ig = spy
            rel, ab, altitudes, entropies, rmses = ig.execute()
        n = params["experiment"]["missions"]["n_agents"]
        T = params["experiment"]["constraints"]["budget"] + 1
        packed = [np.packbits(c) for c in rec.correctness]
        save(tag, episode=np.array(episode), relative_return=np.array(rel), absolute_return=np.array(ab),
             altitudes=np.array(altitudes, dtype=np.int32), entropies=np.array(entropies, dtype=np.float64),
             f1=np.array(rmses, dtype=np.float64), gains=np.array(gains).reshape(T, n, -1),
             corr_packed=np.concatenate(packed), corr_lens=np.array([len(c) for c in rec.correctness], dtype=np.int32),
             comm_draws=np.array(rec.comm))


# ------------------------------------------------------------------ deployment / comparison scripts
def gen_missions():
    """random_baseline.py, lawn_mower.py and coma_test.py run as they are, every random draw recorded."""
    import marl_framework.random_baseline as ref_rb
    import marl_framework.lawn_mower as ref_lm
    import marl_framework.coma_test as ref_ct
    from marl_framework.actor.network import ActorNetwork

    def pack(rec):
        return dict(corr_packed=np.concatenate([np.packbits(c) for c in rec.correctness]),
                    corr_lens=np.array([len(c) for c in rec.correctness], dtype=np.int32))

    p = make_params("small", experiment__missions__n_agents=3)
    torch.manual_seed(5)
    with Recorder() as rec:
        ret, entropies, f1 = ref_rb.RandomBaseline(p, None, 6).execute()
    save("random_small3_e6", episode=np.array(6), ret=np.array(ret), entropies=np.array(entropies), f1=np.array(f1),
         actions=np.array(rec.actions, dtype=np.int32), **pack(rec))

    p = make_params("small", experiment__missions__n_agents=8, experiment__baselines__lawnmower__altitude=10)  # the script needs 8 slots
    torch.manual_seed(6)
    with Recorder() as rec:
        ret, entropies, f1 = ref_lm.LawnMower(p, None, 2).execute()
    save("lawnmower_small_e2", episode=np.array(2), ret=np.array(ret), entropies=np.array(entropies), f1=np.array(f1), **pack(rec))

    p = make_params("small", experiment__missions__n_agents=3)
    torch.manual_seed(31)
    net = ActorNetwork(p)          # the script loads a whole-module pickle from a fixed path: hand it this one instead
    real_load = torch.load
    torch.load = lambda *a, **k: net
    chosen = []
    real_argmax = torch.argmax

    def argmax_spy(x, *a, **k):
        r = real_argmax(x, *a, **k)
        if x.numel() == p["experiment"]["constraints"]["num_actions"]:
            chosen.append(int(r))
        return r

    torch.argmax = argmax_spy
    try:
        torch.manual_seed(32)
        with Recorder() as rec:
            test = ref_ct.COMATest(p, None, 9)
            ret, positions, altitudes, entropies, f1, rel = test.execute("random", 9)
    finally:
        torch.load = real_load
        torch.argmax = real_argmax
    save("comatest_small3_e9", episode=np.array(9), net_seed=np.array(31), ret=np.array(ret), relative_return=np.array(rel),
         positions=np.array(positions, dtype=np.int32), altitudes=np.array(altitudes, dtype=np.int32),
         entropies=np.array(entropies), f1=np.array(f1), actions=np.array(chosen, dtype=np.int32),
         comm_draws=np.array(rec.comm), **pack(rec))


GENERATORS = [gen_derived_and_footprints, gen_start_states, gen_truth, gen_terrain, gen_masks, gen_comm, gen_bayes_measurement,
              gen_entropy_reward, gen_episodes, gen_episode_default_grid, gen_episode_prior_and_c4, gen_td_lambda, gen_coma_step, gen_ig_baseline, gen_missions]

if __name__ == "__main__":
    check_schema()
    only = sys.argv[1:]
    for g in GENERATORS:
        if only and g.__name__ not in only:
            continue
        g()
