"""CPU-only checks of the product's host side: derived constants against the reference's golden vectors, the
C-ABI library (loads, exports every symbol of include/ippmarl.h, struct mirror), and the host-callable mirrors
of the device's integer streams (MT19937 start states / truth parameters, Philox, area-resize weights)."""
import os
import re

import numpy as np
import pytest

import ipp_oracle as O
from configs import make_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _derived(name, **over):
    from ippmarl.derived import DerivedConstants
    return DerivedConstants(make_params(name, **over))


@pytest.mark.parametrize("name", ["default", "small", "c2", "c4", "c5"])
def test_derived_constants_match_reference(golden, name):
    fx = golden("derived_footprints")
    d = _derived(name)
    assert [d.res_x, d.res_y] == list(fx[f"{name}_res"])
    assert [d.grid_x, d.grid_y] == list(fx[f"{name}_dims"])
    assert [d.space_x, d.space_y, d.space_z] == list(fx[f"{name}_space"])
    full, clip = [], []
    for x in range(d.space_x):
        for y in range(d.space_y):
            for z in range(d.space_z):
                f, c = d.footprint(d.index_to_position([x, y, z]))
                full.append(f), clip.append(c)
    assert np.array_equal(np.array(full), fx[f"{name}_fp_full"])  # integer tables reproduce the float64 knife edges
    assert np.array_equal(np.array(clip), fx[f"{name}_fp_clip"])
    assert d.tile_stride % 4 == 0 and d.tile_stride >= 2 * max(d.radius_y) + 3


def test_measurement_tables_match_reference_arithmetic(golden):
    d = _derived("c2")
    fx = golden("bayes_measurement")
    for k, alt in enumerate((5, 10, 15)):
        vals = np.unique(fx[f"meas_out_{alt}"])
        assert list(vals) == sorted(d.meas_value[k])          # float32 values the reference produces
        y = np.float32(vals)
        np.testing.assert_array_equal(np.sort(d.logit_meas[k]), np.log(y / (1 - y)))  # float32 logits, same ops
        assert int(d.flip_threshold[k]) == O.philox_flip_threshold(O.noise_of_altitude(alt))
    assert abs(d.logit_clip - np.log(0.9999 / 0.0001)) < 1e-9


def test_config_validation():
    from ippmarl.derived import DerivedConstants
    p = make_params("default")
    del p["environment"]["x_dim"]
    with pytest.raises((ValueError, KeyError)):
        DerivedConstants(p)
    with pytest.raises(ValueError):
        DerivedConstants(make_params("default", experiment__missions__n_agents=17))
    # the reference starts every UAV at 15 m (agent/state_space.py:32): altitude bounds that leave that level out would fly
    # the UAVs at an altitude the footprint / noise tables do not hold
    with pytest.raises(ValueError, match="15 m"):
        DerivedConstants(make_params("default", experiment__constraints__min_altitude=10, experiment__constraints__max_altitude=10))
    DerivedConstants(make_params("default", experiment__constraints__min_altitude=15, experiment__constraints__max_altitude=15))
    DerivedConstants(make_params("default", experiment__constraints__min_altitude=10, experiment__constraints__max_altitude=20))


def test_random_configurations_are_valid_on_the_host():
    """Every configuration the parity sweep can draw (tests/random_configs.py) passes the host-side validation and stays inside
    the compiled limits -- the sweep itself needs the GPU, this keeps its generator honest on CPU."""
    import random
    from ippmarl.derived import DerivedConstants
    from random_configs import random_case
    rng = random.Random(99)
    seen = set()
    for _ in range(300):
        name, over, n_envs, seed, ep0, n, A = random_case(rng)
        d = DerivedConstants(make_params(name, **over), philox_seed=seed)
        assert d.n_agents == n and d.n_actions == A and 1 <= n_envs <= 48 and 15 in d.altitudes
        assert ep0 * d.env_seed * max(n - 1, 1) < 2 ** 32           # NumPy legacy seeding limit of reset()
        seen.add((name, d.grid_x, A, d.prior != 0.5, max(d.altitudes) > 15))
    assert len(seen) > 40    # grid sizes x action sets x prior x altitude lattices actually vary


def test_library_loads_and_exports_every_declared_symbol():
    from ippmarl import _ffi
    lib = _ffi.load_library()
    header = open(os.path.join(ROOT, "include", "ippmarl.h")).read()
    declared = set(re.findall(r"\b(ippm_[a-z0-9_]+)\s*\(", header))
    declared -= {"ippm_ctx"}
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ippmarl.h but not exported"
    bound = set(_ffi.PROTOTYPES) | {"ippm_last_error", "ippm_version", "ippm_config_size"}
    assert declared == bound, f"binding and header disagree: {declared ^ bound}"
    assert lib.ippm_version() == 500
    import ctypes
    assert lib.ippm_config_size() == ctypes.sizeof(_ffi.IppmConfig)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` outside a launcher (the driver's command shape) starts N ranks itself: here 2 ranks over gloo
    that rendezvous on a free local port, all-reduce their rank ids and stop before any GPU work."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--dist-backend", "gloo"],
                         capture_output=True, text=True, timeout=240, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout    # only rank 0 speaks
    rec = json.loads(lines[0])
    assert rec["ranks"] == 2 and rec["n_gpus"] == 2 and rec["rank_id_sum"] == 1 and rec["backend"] == "gloo"


def test_eight_ranks_rendezvous_and_only_rank_zero_speaks():
    """The 8-GPU command shape of the driver, on CPU: `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`
    (here with --rendezvous-only over gloo): 8 ranks meet on 127.0.0.1, all-reduce their rank ids (0 + ... + 7 = 28), rank 0 alone
    prints; and a launcher that starts a different number of ranks than --gpus says so instead of running."""
    import json
    import socket
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"

    def run(nproc, gpus):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
                               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(gpus),
                               "--rendezvous-only", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)

    out = run(8, 8)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["ranks"] == 8 and rec["n_gpus"] == 8 and rec["rank_id_sum"] == 28 and rec["backend"] == "gloo"
    bad = run(2, 8)
    assert bad.returncode != 0 and "--gpus 8 but the launcher started 2 ranks" in (bad.stderr + bad.stdout)


def test_launcher_retries_once_with_the_other_ipc_setting(monkeypatch):
    """bench.spawn_ranks: the ranks are started with the environment's HSA_ENABLE_IPC_MODE_LEGACY; if they fail, once more with
    the other setting; every attempt gets a fresh port and tells the ranks which setting it is."""
    import bench
    calls = []

    def fake_call(cmd, env):
        calls.append((cmd, env.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset"), env["IPPM_BENCH_IPC_SETTING"]))
        return 1 if len(calls) == 1 else 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert bench.spawn_ranks(8) == 0
    assert [c[1] for c in calls] == ["0", "unset"] and "attempt 2" in calls[1][2]
    ports = [c[0][c[0].index("--master-port") + 1] for c in calls]
    assert all("--nproc-per-node=8" in c[0] for c in calls) and all(p.isdigit() for p in ports)
    calls.clear()
    assert bench.spawn_ranks(4, ipc_mode="unset", retry=False) == 1 and [c[1] for c in calls] == ["unset"]
    calls.clear()
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY")
    bench.spawn_ranks(2)
    assert [c[1] for c in calls] == ["unset", "0"]


def test_fusion_rotation_rule_spreads_every_env_over_all_xcds():
    """The launch rule of the tile fusion and of the row walker's work-list form (fuse_tiles.hip, fuse.hip), restated: workgroup
    (x, f) of a grid (E, per_env) is linear workgroup x + E * f and runs on XCD (x + E * f) % 8; it serves env
    (x + ((f >> s) & 7)) mod E for even E >= 8 (s = 2 where the launch has at least 32 wavefronts per env, else 1 or 0) and env x
    otherwise.  For every batch size: each (env, f) is served exactly once, and an env's wavefronts are spread over the eight XCDs
    when E is odd or rotated -- the property whose absence cost config 5's mixed batches 20-45 % of the fusion."""
    for E in list(range(1, 41)) + [64, 85, 86, 128, 256, 512, 1024]:
        for per_env in (8, 16, 35, 64, 1024):
            rot = 0
            if E >= 8 and E % 2 == 0:
                rot = 1
                while rot < 3 and (per_env >> rot) >= 8:
                    rot += 1
            share = {}
            for f in range(per_env):
                served = set()
                for x in range(E):
                    env = x + (((f >> (rot - 1)) & 7) if rot else 0)
                    env -= E if env >= E else 0
                    served.add(env)
                    per_xcd = share.setdefault(env, [0] * 8)
                    per_xcd[(x + E * f) % 8] += 1
                assert served == set(range(E)), (E, per_env, f)
            if E % 2 == 1 or rot:
                # no XCD holds more than 26 % of any env's wavefronts (unrotated even batches: up to all of them); with 16 or more
                # wavefronts per env every env visits all eight (with 8, the wrap at the batch's end costs the first envs an XCD or two)
                assert max(max(v) for v in share.values()) <= 0.26 * per_env + 1e-9, (E, per_env, rot)
                if per_env >= 16:
                    assert all(min(v) > 0 for v in share.values()), (E, per_env, rot)

def test_split_env_swaps_streams_that_share_a_queue(monkeypatch):
    """SplitVecEnv._spread_streams (vec_env.py) with the device replaced by a model of it: streams are numbered as the pool hands
    them out, stream i is served by hardware queue i % 4, and two streams on one queue take twice as long as side by side.  The parts'
    streams must end up on pairwise different queues, with as few swaps as that takes."""
    torch = pytest.importorskip("torch")
    from ippmarl import vec_env

    class FakeStream:
        count = 0

        def __init__(self, device=None):
            self.idx = FakeStream.count
            FakeStream.count += 1

    def side_by_side(a, b, cycles):
        return (1.0 if a.idx % 4 == b.idx % 4 else 0.53), 1e-3

    monkeypatch.setattr(vec_env.torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(vec_env.torch.cuda, "_sleep", lambda cycles: None, raising=False)
    monkeypatch.setattr(vec_env.SplitVecEnv, "_side_by_side", staticmethod(side_by_side))
    for first, parts, want_redraws in ((0, 2, 0), (0, 3, 0), (3, 2, 0), (0, 4, 0)):
        FakeStream.count = first
        env = object.__new__(vec_env.SplitVecEnv)
        env.device = "cpu"
        env.streams = [FakeStream() for _ in range(parts)]
        redraws, ratios = env._spread_streams()
        assert redraws == want_redraws and max(ratios) < 0.8
        assert len({st.idx % 4 for st in env.streams}) == parts
    # the pool hands out 0, 4 (the same queue): the second is swapped once, for 5
    env = object.__new__(vec_env.SplitVecEnv)
    env.device = "cpu"
    a, b = FakeStream(), FakeStream()
    a.idx, b.idx, FakeStream.count = 0, 4, 5
    env.streams = [a, b]
    assert env._spread_streams() == (1, [0.53]) and [st.idx for st in env.streams] == [0, 5]
    # three parts on queues 0, 1, 1: the third moves on until it is beside both
    env = object.__new__(vec_env.SplitVecEnv)
    env.device = "cpu"
    env.streams = [FakeStream(), FakeStream(), FakeStream()]
    env.streams[0].idx, env.streams[1].idx, env.streams[2].idx, FakeStream.count = 0, 1, 5, 8
    redraws, ratios = env._spread_streams()
    assert redraws == 3 and [st.idx for st in env.streams] == [0, 1, 10] and ratios == [0.53, 0.53]   # 8 -> queue 0, 9 -> 1, 10 -> 2


def test_placement_search_stop_rule():
    """VecEnv.tune_placement's early exit: stops on a clear fast draw whether fast draws are the minority or the
    majority, never on a slow outlier among slow draws, and gives up on a box with one kind only after twelve draws."""
    from ippmarl.vec_env import placement_stop_reason as stop
    assert stop([121.5]) is None and stop([121.5, 122.0]) is None
    assert stop([121.5, 122.0, 113.4]) == "a fast allocation found"            # slow, slow, fast
    assert stop([113.4, 121.9]) is None                                         # one of each: which is the outlier?
    assert stop([113.4, 121.9, 113.9]) is None                                  # fast, slow, fast looks exactly like slow, outlier, slow
    assert stop([113.4, 121.9, 113.9, 122.4]) == "a fast allocation found"     # both kinds seen twice
    assert stop([113.4, 113.9, 114.1, 122.3]) is None                           # fast draws in the majority: one slow draw may be an outlier
    assert stop([113.4, 113.9, 114.1, 122.3, 121.8]) == "a fast allocation found"   # ... two that agree are the other kind
    assert stop([121.0, 121.2, 126.0]) is None                                  # a slow outlier among slow draws
    assert stop([121.0, 121.5, 129.0]) is None                                  # (more than 6 % above, still one outlier)
    assert stop([121.0, 121.5, 129.0, 134.0]) is None                           # two outliers that do not agree are not a kind either
    assert stop([121.0, 121.2, 126.0, 121.4, 120.9]) is None
    assert stop([121.0 + 0.1 * k for k in range(11)]) is None
    assert stop([121.0 + 0.1 * k for k in range(12)]) == "no spread between the first draws"


def test_placement_search_memory_bound():
    """VecEnv.tune_placement keeps rejected candidates allocated only within half of the free device memory."""
    from ippmarl.vec_env import placement_alive_cap
    GB = 1 << 30
    assert placement_alive_cap(280 * GB, int(1.7 * GB)) >= 24          # config 2 on an empty MI355X: the whole search fits
    assert placement_alive_cap(280 * GB, int(9.7 * GB)) == 14          # config 4's per-GPU shape: 13 candidates + the arena in use
    assert placement_alive_cap(30 * GB, int(9.7 * GB)) == 2            # next to a trainer's activations: one candidate at a time
    assert placement_alive_cap(0, GB) == 2


def test_missing_library_fails_loudly(monkeypatch):
    from ippmarl import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", "/nonexistent/libippmarl.so")
    with pytest.raises(_ffi.IppmError, match="no CPU fallback"):
        _ffi.load_library()


def test_no_gpu_means_no_env():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ippmarl import _ffi
    from ippmarl.vec_env import VecEnv
    with pytest.raises(_ffi.IppmError):
        VecEnv(make_params("small"), 2)


def test_device_mt19937_mirror_matches_numpy(golden):
    from ippmarl import _ffi
    fx = golden("start_states")
    got = np.array([[_ffi.host_start_state(3, e, a, 5, 11, 11) for a in range(16)] for e in range(1, 65)])
    assert np.array_equal(got, fx["seed3"])
    got7 = np.array([[_ffi.host_start_state(7, e, a, 5, 11, 11) for a in range(4)] for e in range(1, 17)])
    assert np.array_equal(got7, fx["seed7"])
    tp = golden("truth")["split_pct"]
    assert np.array_equal(np.array([_ffi.host_truth_params(e) for e in range(1, 4097)]), tp)
    # large episode numbers (later reset waves)
    d = O.Derived(make_params("default"))
    for e in (10 ** 6 + 7, 123456789):
        assert _ffi.host_truth_params(e) == O.truth_split_params(e)
        assert _ffi.host_start_state(3, e, 3, 5, 11, 11) == list(O.start_state(d, 3, e))


def test_philox_mirror_matches_oracle_and_known_answers():
    from ippmarl import _ffi
    assert _ffi.host_philox(0, 0, 0, 0, 0, 0) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert _ffi.host_philox(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    rng = np.random.RandomState(0)
    for _ in range(50):
        args = [int(v) for v in rng.randint(0, 2 ** 32, size=6, dtype=np.uint64)]
        assert _ffi.host_philox(*args) == [int(v) for v in O.philox4x32(*args)]


@pytest.mark.parametrize("n_src", [30, 60, 90, 128, 256, 493, 512, 1024])
def test_area_weights_match_oracle(n_src):
    from ippmarl import _ffi
    b, w0, w1 = _ffi.host_area_weights(n_src)
    dense = O.area_weights(n_src, 11)
    mine = np.zeros_like(dense)
    for i in range(n_src):
        mine[b[i], i] += w0[i]
        if b[i] + 1 < 11:
            mine[b[i] + 1, i] += w1[i]
    np.testing.assert_allclose(mine, dense, rtol=0, atol=6e-8)  # weights are handed over as float32
    np.testing.assert_allclose(mine.sum(axis=1), 1.0, atol=1e-6)


def test_params_schema_loads_and_matches_reference_defaults():
    from ippmarl.params import load_params, grid256_params
    p = load_params()
    assert p["experiment"]["constraints"]["budget"] == 14 and p["networks"]["lambda"] == 0.8
    assert p["experiment"]["uav"]["communication_range"] == 25 and p["mapping"]["prior"] == 0.5
    from ippmarl.derived import DerivedConstants
    assert DerivedConstants(grid256_params()).grid_x == 256


def test_hot_plane_layout_is_aligned_and_disjoint():
    """VecEnv keeps its large planes in one allocation: 2 MB-aligned, non-overlapping spans, config 2's sizes."""
    import torch
    from ippmarl.vec_env import hot_layout
    shapes = (("local", (1024, 4, 256, 256), torch.float32), ("glob", (1024, 256, 256), torch.float32),
              ("code", (1024, 4, 96 * 96 // 4), torch.uint8), ("truth", (1024, 8192), torch.uint8), ("odd", (3, 5, 7), torch.float32))
    spans, total = hot_layout(shapes)
    assert [n for _, n in spans] == [1 << 30, 1 << 28, 1024 * 4 * 2304, 1 << 23, 420]
    end = 0
    for off, n in spans:
        assert off % (2 << 20) == 0 and off >= end
        end = off + n
    assert total % (2 << 20) == 0 and total >= end and total - end < (2 << 20)


def test_tools_and_gpu_scripts_parse():
    """tools/ holds the round's probes and gpurun scripts (not product code, none of it imported by the package): every Python file
    there compiles and every shell script passes `bash -n`, so that a call on the GPU box does not die on a typo."""
    import glob
    import subprocess
    tools = os.path.join(ROOT, "tools")
    scripts = sorted(glob.glob(os.path.join(tools, "*.py")))
    assert len(scripts) > 30
    for path in scripts:
        compile(open(path).read(), path, "exec")
    for path in sorted(glob.glob(os.path.join(tools, "*.sh"))):
        assert subprocess.run(["bash", "-n", path]).returncode == 0, path
    # nothing under tools/ or oracle/ is imported by the product package
    for path in glob.glob(os.path.join(ROOT, "ipp-marl_amd", "ippmarl", "**", "*.py"), recursive=True):
        text = open(path).read()
        assert "import oracle" not in text and "from oracle" not in text and "ipp_oracle" not in text.replace("oracle/ipp_oracle.py", ""), path
        assert "from tools" not in text and "import tools" not in text, path

