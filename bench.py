#!/usr/bin/env python
"""bench.py -- agent-env steps/s of the batched multi-UAV environment step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 150 --warmup 30
    python bench.py --gpus 8 --steps 150 --warmup 30        # starts its own 8 ranks (one per GPU, RCCL) when not under a launcher
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 150 --warmup 30           # the same, under an external launcher

A "step" is one environment step of ALL envs on a rank, three launches per sub-batch: comm matrix + fusion plans + mask/act/move
with a uniform random valid policy (K1) -> local fusion (K4) and global fusion + reward terms (K5) -> sense + Bayes update at the
new positions (K3, which also completes the reward); every 15 steps the envs are reset to fresh episodes inside the timed
region (device-side MT19937 + Philox, terrain synthesis included).  The batch is stepped as --streams sub-batches, each on its own
HIP stream (default 2: the latency-bound plan kernel and the reset of one half run beside the bandwidth-bound map kernels of the
other; same episodes, bit for bit); the roofline leg then measures every kernel with the whole batch per launch, alone on the device.
Default workload = BASELINE.json configs[1]: 4 UAVs, 256 x 256 grid, 1024 batched envs per GPU, random policy.
Inputs are synthetic and resident in HBM (truth fields generated on the device).  Envs are independent, so N GPUs
shard envs with no data-path collective ("weak" scaling: 1024 envs per GPU).

The JSON line also carries
  roofline         : the map-update kernel K3 (k_sense_tiles): algorithmic bytes = 10 B per footprint cell (4R+4W posterior,
                     1R truth, 1W measurement code) / the kernel's own begin-to-end duration, taken from HIP
                     events bound to each dispatch (hipExtLaunchKernelGGL; what rocprofv3's kernel trace reports) over a
                     separate leg of --roofline-steps env steps run right after the timed region -- so the figure does not
                     depend on --steps and the timed region carries no events at all;  whole_step = all algorithmic bytes of a
                     step / ms_per_step / peak
  roofline_kernels : the same for the fusion launch (K4+K5: 8 B per cell of the union + 1 B per (cell, message)), the small
                     plan kernel, the reset kernels and, when training is on, the K6 feature builders
  resets_timed, steady_state : how many episode resets the timed window held, and the same loop's rate over --steady-episodes (20)
                     whole episodes right after it (exactly one reset per episode): the figure to quote for sustained throughput
                     (also under roofline.steady_state, which the driver's record keeps)
  roofline.batch_leg, config.map_layout : the same shape and loop at TWICE the batch (2048 envs by default; ten whole episodes; one GPU only), where
                     map_layout="auto" stores the maps as 128-byte tiles (ippm_set_map_layout: row-major rows get dearer per cell as a launch's maps
                     grow past ~2 GB, whole lines do not) -- NOT the metric's configuration, never `value`; and the layout the timed envs used
  per_rank         : every rank's own ms_per_step / rate / placement-search outcome; value_sum_of_ranks next to value_from_max_time
  ranks, rank_devices, collective : who took part (one entry per rank) and the gradient all-reduces RCCL carried in the
                     COMA leg (backend, calls, bytes)
  cpu_baseline     : the NumPy oracle (a port of the reference's CPU path, parity-pinned against it) stepping the same
                     config on one host core (one env) and on all host cores (64 envs), bounded to ~10 s each; the
                     reference itself as probed in the build container is quoted as reference_probe
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
K3_BYTES_PER_CELL = 10
# what the kernels' own layout moves per footprint cell: 4 R + 4 W belief, truth bit-packed (1/8 B), codes one byte per 4-cell group
K3_LAYOUT_BYTES_PER_CELL = 8.375
# the unmodified reference, measured in the build container on 8 host cores; it cannot travel to the GPU box
REFERENCE_PROBE = {"agent_env_steps_per_s": 10.0, "coma_updates_per_s": 0.164, "cores": 8,
                   "source": "SURVEY.md section 6: /root/reference run in the build container (default params, 493x493)"}
PIXELS = {128: 15, 256: 30, 512: 60, 1024: 120}


def bench_params(args):
    from ippmarl.params import grid256_params
    number = PIXELS[args.grid]   # other BASELINE.json grid sizes are parity-test configs; 256 is the metric's
    over = dict(experiment__missions__n_agents=args.agents, sensor__pixel__number_x=number, sensor__pixel__number_y=number)
    if getattr(args, "actions", None):            # BASELINE config 5's shape: 27 actions (3-D moves), per-episode comm range
        over["experiment__constraints__num_actions"] = args.actions
    if getattr(args, "episode_comm_range", False):
        over["experiment__uav__fix_range"] = False
    if getattr(args, "comm_range", None) is not None:
        over["experiment__uav__communication_range"] = args.comm_range
    return grid256_params(**over)


def cpu_baseline(args, seconds=10.0):
    """Oracle (kind 'port') on host cores, in separate processes (oracle/ never enters this process): one env on one core,
    then 64 envs spread over all cores."""
    script = os.path.join(ROOT, "oracle", "cpu_bench.py")
    common = ["--agents", str(args.agents), "--number", str(PIXELS[args.grid]), "--terrain", args.terrain, "--seconds", str(seconds)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")

    def launch(n_envs, first):
        return subprocess.Popen([sys.executable, script, "--envs", str(n_envs), "--first-episode", str(first)] + common,
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)

    def collect(procs):
        steps, dt = 0, 0.0
        for p in procs:
            out, _ = p.communicate(timeout=seconds * 6 + 60)
            rec = json.loads(out.strip().splitlines()[-1])
            steps += rec["agent_env_steps"]
            dt = max(dt, rec["seconds"])
        return steps, dt

    cores = os.cpu_count() or 1
    steps1, dt1 = collect([launch(1, 1)])
    procs = min(cores, 64)
    per = -(-64 // procs)
    stepsN, dtN = collect([launch(per, 1 + k * 100000) for k in range(procs)])
    shape = f"{args.agents} UAVs, {args.grid}x{args.grid}, env-only, {args.terrain} terrain"
    return {"value": steps1 / dt1, "unit": "agent-env steps/s", "cores": 1, "kind": "port", "host_cores": cores,
            "torch_threads": torch.get_num_threads(), "numpy_threads": 1,
            "form": "the oracle steps ONE env at a time, like the reference (missions/episode_generator.py:39-40); it has no form vectorised "
                    "over envs, so the many-env figure is processes x one env each, not SURVEY 8d's E=64 vectorised stepper -- which would not "
                    "read differently per core: one env step of the oracle already is ~16 full-grid NumPy passes (every fusion clips and updates "
                    "all 65 536 cells, mappings.py:109-124) of ~0.4 ms each, so batching 64 envs into each pass multiplies cells and time alike",
            "sample": f"{steps1} agent-env steps of 1 env ({shape}) in {dt1:.1f}s of the NumPy oracle on 1 of {cores} host cores",
            "many_env": {"value": stepsN / dtN, "unit": "agent-env steps/s", "cores": procs, "envs": per * procs,
                         "sample": f"{stepsN} agent-env steps of {per * procs} envs in {dtN:.1f}s, {procs} oracle processes "
                                   f"({per} envs each, one core each)"},
            "reference_probe": REFERENCE_PROBE}


def dropin_seam(device, episodes=3):
    """Throughput of the drop-in seam itself: EpisodeGenerator.execute over the reference-shaped objects (COMAWrapper / Agent /
    Mapping / CommunicationLog / BatchMemory: one env, NumPy at the boundary, the calls of missions/episode_generator.py:38-88 and
    coma_wrapper.py:73-183 one by one), which is what a user who only swaps the imports of INTEGRATION.md section A runs.  Timed
    for BASELINE config 2's parameters and for the reference's default 493 x 493 grid; with the host-side breakdown per agent-env
    step: calls into libippmarl.so and device-to-host reads (Tensor.cpu / item / tolist / int() / float() on device tensors)."""
    import numpy as np
    from ippmarl import _ffi
    from ippmarl.batch_memory import BatchMemory
    from ippmarl.coma_wrapper import COMAWrapper
    from ippmarl.mapping.grid_maps import GridMap
    from ippmarl.missions.episode_generator import EpisodeGenerator
    from ippmarl.params import default_params, grid256_params
    from ippmarl.sensors import Sensor
    from ippmarl.sensors.models import SensorModel
    count = {"lib": 0, "d2h": 0}
    real_call = _ffi.Context.call

    def counted_call(self, name, *a):
        count["lib"] += 1
        return real_call(self, name, *a)

    readers = ("cpu", "item", "tolist", "__int__", "__float__", "__bool__", "__index__")
    real = {k: getattr(torch.Tensor, k) for k in readers}

    def counting(k):
        def f(self, *a, **kw):
            if self.is_cuda:
                count["d2h"] += 1
            return real[k](self, *a, **kw)
        return f

    out = {}
    _ffi.Context.call = counted_call
    for k in readers:
        setattr(torch.Tensor, k, counting(k))
    try:
        for tag, params in (("config2_256x256", grid256_params(experiment__missions__n_agents=4)), ("default_493x493", default_params())):
            n = params["experiment"]["missions"]["n_agents"]
            T = params["experiment"]["constraints"]["budget"] + 1
            np.random.seed(7)
            torch.manual_seed(7)
            wrapper = COMAWrapper(params, None, device=device)
            grid_map = GridMap(params)
            gen = EpisodeGenerator(params, None, grid_map, Sensor(SensorModel(), grid_map))
            gen.execute(1, BatchMemory(params, wrapper), wrapper, "train")      # warm-up: library / MIOpen first use
            torch.cuda.synchronize()
            count["lib"] = count["d2h"] = 0
            t0 = time.perf_counter()
            for ep in range(2, 2 + episodes):
                memory = BatchMemory(params, wrapper)
                gen.execute(ep, memory, wrapper, "train")
                assert memory.size() == T * n
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            steps = episodes * T * n
            out[tag] = {"agent_env_steps_per_s": steps / dt, "s_per_episode": dt / episodes, "episodes": episodes, "n_agents": n,
                        "library_calls_per_agent_step": count["lib"] / steps, "device_to_host_reads_per_agent_step": count["d2h"] / steps}
    finally:
        _ffi.Context.call = real_call
        for k in readers:
            setattr(torch.Tensor, k, real[k])
    out["what"] = ("EpisodeGenerator.execute through the reference-shaped objects, one env, 'train' mode (actor forward at batch 1 per agent, "
                   "transitions into BatchMemory); reference: ~10 agent-env steps/s (SURVEY section 6). For throughput use VecEnv / COMATrainer.")
    return out


def spawn_ranks(n: int, ipc_mode: str = "keep", retry: bool = True) -> int:
    """`python bench.py --gpus N` outside a launcher: re-run this command as N ranks (one per GPU) under
    torch.distributed.run on a free local port.  The ranks' stdout is ours, so rank 0's JSON line comes out as usual."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    # HSA_ENABLE_IPC_MODE_LEGACY: the image this was built on exports 0 ("the host driver only supports dmabuf IPC": without it RCCL
    # fails with hipIpcGetMemHandle: invalid argument).  It is read when the HSA runtime starts, so it cannot be changed inside a
    # rank: the launcher tries the environment's own value first (--ipc-legacy keep; or the value forced by --ipc-legacy 0 / unset)
    # and, if the ranks fail, ONCE more with the other setting, and says on stderr which one worked.
    first = ipc_env_setting(env, ipc_mode)
    order = [first, "unset" if first == "0" else "0"]
    rc = 1
    for attempt, setting in enumerate(order):
        e = dict(env)
        e.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
        if setting != "unset":
            e["HSA_ENABLE_IPC_MODE_LEGACY"] = setting
        e["IPPM_BENCH_IPC_SETTING"] = f"HSA_ENABLE_IPC_MODE_LEGACY={setting} (attempt {attempt + 1})"
        with socket.socket() as sock:   # a fresh port per attempt
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        rc = subprocess.call(cmd, env=e)
        if rc == 0:
            if attempt:
                print(f"bench.py: the ranks ran with HSA_ENABLE_IPC_MODE_LEGACY={setting} after failing with {order[0]}", file=sys.stderr)
            return 0
        print(f"bench.py: {n} ranks failed (exit {rc}) with HSA_ENABLE_IPC_MODE_LEGACY={setting}"
              + ("; retrying once with the other setting" if attempt == 0 and retry else ""), file=sys.stderr)
        if not retry:
            break
    return rc


def ipc_env_setting(env, mode: str) -> str:
    """The first HSA_ENABLE_IPC_MODE_LEGACY setting the launcher tries: "keep" = whatever the environment holds."""
    if mode == "keep":
        return env.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset")
    return mode


def newest_pmc_summary(n_envs, n_agents, grid, launch_envs=None):
    """profiles/rNN/pmc_summary*.json of the newest round that has one FOR THIS SHAPE (written by tools/pmc_summary.py from
    separate rocprofv3 --pmc passes of this command: pmc_summary.json = config 2, pmc_summary_c4.json / _c5.json = the shapes of
    BASELINE configs 4 and 5)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_summary*.json")), reverse=True):
        with open(path) as f:
            rec = json.load(f)
        if (rec.get("envs_per_gpu"), rec.get("n_agents"), rec.get("grid")) == (n_envs, n_agents, grid) and \
                rec.get("envs_per_launch", rec.get("envs_per_gpu")) == (launch_envs or n_envs):
            return rec, os.path.relpath(path, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--grid", type=int, default=256, choices=sorted(PIXELS))
    ap.add_argument("--actions", type=int, default=None, choices=[4, 6, 9, 27], help="action set (default: params.yaml's 6)")
    ap.add_argument("--episode-comm-range", action="store_true", help="per-episode comm range from {0, 15, 25, 100} m "
                    "(experiment.uav.fix_range: False, BASELINE config 5's comm-range masking)")
    ap.add_argument("--streams", type=int, default=0, help="step the batch as this many sub-batches, each on its own HIP stream "
                    "(ippmarl.vec_env.SplitVecEnv): the latency-bound plan kernel and the reset of one runs beside the bandwidth-bound map "
                    "kernels of the other -- 0.1526 -> 0.141 ms per step at config 2; 1 = one stream, one launch per kernel and step; "
                    "0 (default) = 2, or 3 where the envs' work differs many-fold (--episode-comm-range, --team-sizes: 1.09 / 1.13 / 1.32 M "
                    "agent-env steps/s on 1 / 2 / 3 streams at config 5's shape)")
    ap.add_argument("--stagger", action="store_true", help="every sub-batch in its own phase of the episode (part k runs k * T / parts steps ahead, "
                    "so that at most one part resets at any step) instead of in lock step.  Off: measured in round 6 it gains nothing -- steady state "
                    "0.1386-0.1394 ms per step staggered against 0.1369-0.1382 in lock step, alternating processes on one box "
                    "(profiles/r06/stagger_ab.txt) -- the resets of two halves side by side cost no more than one after the other")
    ap.add_argument("--team-sizes", default=None, help="comma-separated team sizes dealt out to the envs in turn (BASELINE config 5's mixed teams, "
                    "e.g. 2,4,8,16 with --agents 16): env e flies team_sizes[e %% len] of the --agents UAVs; agent-env steps count the flying ones")
    ap.add_argument("--comm-range", type=float, default=None, help="experiment.uav.communication_range in metres (default: params.yaml's 25)")
    ap.add_argument("--terrain", default="random_field", choices=["random_field", "split"],
                    help="ground truth: the power-law random field of ground_truths.py:25-40 generated on the device, or "
                         "the half-plane split the reference flies over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--terrain-prefetch", action="store_true", help="synthesise each wave's terrain beside the previous wave's steps "
                    "on a side stream (VecEnv.prefetch_terrain) instead of inside its reset.  Off: measured 0.176 vs 0.174 ms per "
                    "step -- the step kernels fill the device, the synthesis only takes CUs from them (profiles/r04)")
    ap.add_argument("--roofline-steps", type=int, default=90, help="env steps (resets included) of the roofline leg that follows "
                    "the timed region: every K3 / fusion / plan / reset launch of it carries start/stop events bound to the "
                    "dispatch itself; 0 = no roofline leg")
    ap.add_argument("--train-rounds", type=int, default=2, help="COMA rounds (rollout with the actor + full update) timed after "
                    "the env-only region for the COMA updates/s figure; 0 disables")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo lets the "
                    "multi-rank code path be exercised on a single GPU)")
    ap.add_argument("--graphs", type=int, default=0, help="replay {plan+K1, K4+K5} of each step from a hipGraph; K3 stays an "
                    "ordinary launch.  Off by default: three launches per step do not need it")
    ap.add_argument("--rendezvous-only", action="store_true", help="start the ranks, all-reduce the rank ids over --dist-backend, "
                    "print {\"ranks\": N, ...} and stop before any GPU work (the CPU test of the self-launch)")
    ap.add_argument("--placement-draws", type=int, default=24, help="allocations of the env's hot planes tried before the run "
                    "(VecEnv.tune_placement; 1 = take what the allocator hands out)")
    ap.add_argument("--ipc-legacy", default="keep", choices=["keep", "0", "1", "unset"], help="HSA_ENABLE_IPC_MODE_LEGACY the ranks "
                    "started by `--gpus N` run under: keep = the environment's own value (this image exports 0: the host driver only "
                    "supports dmabuf IPC); if the ranks fail they are started ONCE more with the other setting (--no-ipc-retry: not)")
    ap.add_argument("--no-ipc-retry", dest="ipc_retry", action="store_false")
    ap.add_argument("--steady-episodes", type=int, default=20, help="whole episodes of the steady-state leg that follows the timed region "
                    "(exactly one reset per episode; 20 episodes = 300 steps, ~45 ms at config 2)")
    ap.add_argument("--no-dropin-seam", action="store_true", help="skip the timing of the drop-in object surface (EpisodeGenerator.execute, one env)")
    ap.add_argument("--no-batch-leg", dest="batch_leg", action="store_false", help="skip the leg that steps the same shape at twice the batch "
                    "(roofline.batch_leg: where tile storage of the maps takes over from row-major rows)")
    ap.add_argument("--calib", action="store_true", help="PMC calibration: 3 device-to-device copies of the local maps (known "
                    "bytes read and written by a 16 B/lane streaming kernel) before the timed loop")
    args = ap.parse_args()
    if args.streams <= 0:
        args.streams = 3 if (args.episode_comm_range or args.team_sizes) else 2

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, args.ipc_legacy, args.ipc_retry))   # not under a launcher: start the ranks ourselves and relay rank 0's line
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE)")
    dist = None
    placement_note = None
    if args.rendezvous_only:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.dist_backend == "nccl" and not torch.cuda.is_available() else args.dist_backend)
        ids = torch.tensor([rank], dtype=torch.int64)
        dist.all_reduce(ids)
        if rank == 0:
            print(json.dumps({"ranks": dist.get_world_size(), "n_gpus": world, "rank_id_sum": int(ids[0]),
                              "backend": dist.get_backend(), "launcher_ipc_setting": os.environ.get("IPPM_BENCH_IPC_SETTING")}), flush=True)
        dist.destroy_process_group()
        return
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n_dev = torch.cuda.device_count()
        if args.dist_backend == "nccl":
            if n_dev < world:
                raise SystemExit(f"--gpus {world} with RCCL needs {world} visible GPUs, found {n_dev} "
                                 "(--dist-backend gloo lets the ranks share a GPU for a dry run of the control flow)")
            try:
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
                probe = torch.ones(1, device=f"cuda:{local_rank}")
                dist.all_reduce(probe)          # the first collective is where an IPC problem shows (communicator set-up is lazy)
                torch.cuda.synchronize()
                assert int(probe[0]) == world
            except Exception as exc:   # noqa: BLE001
                raise SystemExit(f"rank {rank}: RCCL did not come up ({type(exc).__name__}: {exc}); HSA_ENABLE_IPC_MODE_LEGACY="
                                 f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset')} -- it is read when the HSA runtime starts: "
                                 "relaunch with the other setting (`python bench.py --gpus N` does that by itself, once)")
        else:
            dist.init_process_group(args.dist_backend)
        if world > n_dev and args.placement_draws > 1:
            # ranks share a device (a gloo dry run of the control flow): their placement searches would time each other's kernels
            # and hold candidate arenas side by side on one card -- no search, the allocator's first arena is kept
            args.placement_draws = 1
            placement_note = f"skipped: {world} ranks share {n_dev} device(s)"
        local_rank %= n_dev
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(device)

    from ippmarl import _ffi
    from ippmarl.parallel import episode_ids
    from ippmarl.vec_env import SplitVecEnv, VecEnv, POLICY_UNIFORM
    params = bench_params(args)
    # env-only stepping never builds network inputs: the area sums are not tracked here (the trainer below tracks them)
    teams = None
    if args.team_sizes:
        pattern = [int(v) for v in args.team_sizes.split(",")]
        teams = [pattern[e % len(pattern)] for e in range(args.envs)]
    split = args.streams > 1 and not args.graphs and not args.terrain_prefetch
    if split:   # the batch as sub-batches on their own streams (same envs, same episodes, same results)
        env = SplitVecEnv(params, args.envs, parts=args.streams, device=device, philox_seed=3, terrain=args.terrain, team_sizes=teams)
        # (streams that shared a hardware queue with an earlier one and were swapped; the last side-by-side ratios measured)
        stream_check = {"redraws": env.stream_redraws, "side_by_side": env.stream_probe, "error": env.stream_check_error}
    else:
        stream_check = None
        env = VecEnv(params, args.envs, device=device, philox_seed=3, terrain=args.terrain, track_area=False, team_sizes=teams)
    # The roofline leg measures the kernels ONE LAUNCH AT A TIME at the full batch: with sub-batches on several streams that is a
    # second, whole-batch VecEnv (same config, same library, its own placement search), stepped on one stream after the timed loops.
    map_layout = "tiles: 128-byte tiles of 4 rows x 8 cells (ippm_set_map_layout)" if env.tiled else "rows: row-major [grid_x, grid_y]"
    roof_env = VecEnv(params, args.envs, device=device, philox_seed=3, terrain=args.terrain, track_area=False, team_sizes=teams) \
        if split and args.roofline_steps > 0 else env
    first = roof_env if split and args.roofline_steps > 0 else (env.parts[0] if split else env)   # (copy-rate yardstick, PMC calibration)
    # where the allocator puts the maps is worth 10 % of the fusion kernel (VecEnv.tune_placement): a few candidate sets, one
    # episode each, before anything is timed
    placement = env.tune_placement(args.placement_draws)
    roof_placement = roof_env.tune_placement(args.placement_draws) if roof_env is not env else None
    E, N, T = env.E, env.d.n_agents, env.d.budget + 1
    flying = sum(teams) if teams else E * N        # agents stepped per env step of this rank (mixed team sizes: the flying ones)
    env_actions = env.d.n_actions
    wave = [0]

    def reset():
        env.reset(episode_ids(1, wave[0], E, rank, world))   # disjoint episodes per rank and wave
        wave[0] += 1
        if args.terrain_prefetch:
            # the next wave's terrain is synthesised on a side stream beside this wave's steps (same work, inside the same timed
            # region: every reset of the timed loop is followed by one synthesis); the reset then copies 8 MB of packed truth
            env.prefetch_terrain(episode_ids(1, wave[0], E, rank, world))

    state = {"t": 0}

    def run(n):
        """n env steps of the whole batch, resets included; -> resets of the whole batch among them (a sub-batch's reset counts as
        its share: with staggered sub-batches the parts reset at different steps)."""
        if split:        # SplitVecEnv owns the episode loop: every part in its own phase of the episode
            r0 = env.part_resets
            for _ in range(n):
                env.advance(POLICY_UNIFORM)
            return (env.part_resets - r0) / args.streams
        resets = 0
        for _ in range(n):
            if args.graphs:
                env.step_graphed(state["t"])
            else:
                env.steps(state["t"], policy=POLICY_UNIFORM, features=False)
            state["t"] += 1
            if state["t"] == T:
                resets += 1
                reset()
                state["t"] = 0
        return resets

    if split:
        # part k flies its slice of every wave, all parts in lock step (--stagger: part k starts k * T / parts steps ahead, so that at
        # most one sub-batch resets at any step; the steps taken ahead are part of the untimed start)
        env.start(lambda w: episode_ids(1, w, E, rank, world), stagger=args.stagger)
    else:
        reset()
    if args.graphs:
        env.capture_step_graphs(POLICY_UNIFORM)
    scratch = torch.empty_like(first.local)

    def stream_copy():
        first.ctx.call("ippm_stream_copy", first._p(first.local), _ffi.ptr(scratch), first.local.numel() * 4, first.stream)

    if args.calib:
        torch.cuda.synchronize()     # (the sub-batches' resets run on their own streams: the marker copies must not fall among them)
        for _ in range(3):
            stream_copy()
        torch.cuda.synchronize()
    run(args.warmup)
    env.counters(reset=True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    resets_timed = run(args.steps)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_rank = dt
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    counters = env.counters()
    # Steady state: the timed region above holds however many resets --steps happens to span (the driver's 20 steps: one, i.e. one
    # per 20 steps where an episode has one per 16); any window of whole episodes holds exactly one reset per episode whatever its
    # phase, so the loop simply goes on for --steady-episodes (20) more episodes' worth of steps under the same barrier / max-over-ranks clock.
    ss_steps, ss_resets = args.steady_episodes * T, 0
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    s0 = time.perf_counter()
    ss_resets = run(ss_steps)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ss_dt = time.perf_counter() - s0
    if dist:
        tt = torch.tensor([ss_dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ss_dt = float(tt[0])
    after_ss = env.counters()
    ss_counters = {k: after_ss[k] - counters[k] for k in counters}      # the steady-state leg's own cells
    faults = int(env.fault.abs().sum())        # (SplitVecEnv: joins its streams first)
    grid = [env.d.grid_x, env.d.grid_y]

    # Roofline leg: the same loop goes on for --roofline-steps more env steps (resets included) with kernel timing on -- every
    # launch of the timed classes carries a start/stop event pair bound to its own dispatch (kernel begin -> end, the figure
    # rocprofv3's kernel trace reports; no barrier packets, nothing to calibrate away) -- and with its own work counters, so
    # bytes and durations cover exactly the same launches whatever --steps was.
    times, rl_counters, rl_resets, overlapped = {}, None, 0, None
    if args.roofline_steps > 0 and not args.graphs:
        if split:
            # With sub-batches on several streams a kernel's begin-to-end time in the loops above includes what the other streams'
            # kernels took from it.  First a short stretch of that loop with the events on (`overlapped_us`: what a sub-batch's kernels
            # take there; one stream's plan + fusion + K3 add up to the step); then the roofline leg proper: the WHOLE batch per launch
            # on one stream (roof_env), every launch alone on the device -- the kernel's own rate, the figure rocprofv3 reports for
            # the same launches (tools/loop_stats.py and tools/pmc_summary.py tell the stretches apart by the marker copies of --calib).
            env.profile = True
            run(2 * T)
            env.profile = False
            ov = env.event_times_us()
            overlapped = {k: {"avg_us": v["avg_us"], "launches": v["launches"]} for k, v in ov.items()}
            torch.cuda.synchronize()
        # the leg itself: whole episodes of the whole batch on ONE stream (roof_env is env itself with --streams 1)
        leg_wave = [1 << 16]       # (the leg's own episodes, far from the waves the loops above flew)

        def leg_reset():
            roof_env.reset(episode_ids(1, leg_wave[0], E, rank, world))
            leg_wave[0] += 1

        if roof_env is not env:
            leg_reset()
            for t in range(T):     # untimed: first touch of its arena, one whole episode
                roof_env.steps(t, policy=POLICY_UNIFORM, features=False)
            leg_reset()
            torch.cuda.synchronize()
            if args.calib:       # marker for the profile tools: what follows, up to the yardstick copies, is the roofline leg
                for _ in range(2):
                    stream_copy()
                torch.cuda.synchronize()
        roof_env.counters(reset=True)
        roof_env.profile = True
        if roof_env is env:
            rl_resets = run(args.roofline_steps)
        else:
            t_leg = 0
            for _ in range(args.roofline_steps):
                roof_env.steps(t_leg, policy=POLICY_UNIFORM, features=False)
                t_leg += 1
                if t_leg == T:
                    rl_resets += 1
                    leg_reset()
                    t_leg = 0
        roof_env.profile = False
        times = roof_env.event_times_us()
        rl_counters = roof_env.counters()
    # second denominator: what a plain 16 B/lane device-to-device copy of the local maps reaches on this box
    stream_copy()
    ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ca.record()
    for _ in range(3):
        stream_copy()
    cb.record()
    torch.cuda.synchronize()
    copy_gbs = 3 * 2 * first.local.numel() * 4 / (ca.elapsed_time(cb) * 1e-3) / 1e9
    del scratch
    launch_envs = roof_env.E if args.roofline_steps > 0 else first.E     # envs per kernel launch of the roofline leg
    sub_envs = env.sizes[0] if split else E                              # envs per kernel launch of the timed loops
    pmc, pmc_path = newest_pmc_summary(E, N, grid[0], launch_envs)
    pmc_ok = bool(pmc) and not teams and args.comm_range is None     # (counter passes of this command at this shape and workload)

    TIMING = ("HIP start/stop events bound to each dispatch (hipExtLaunchKernelGGL): the kernel's own begin-to-end duration, as "
              "rocprofv3's kernel trace reports it; no bracket overhead to subtract, so frac_raw == frac")

    def roofline_entry(pmc_key, what, bytes_per_launch, timed, extra=None, layout_bytes=None):
        if timed is None or not timed["launches"]:
            return None
        us = timed["avg_us"]
        achieved = bytes_per_launch / (us * 1e-6) / 1e9
        traffic = pmc.get(pmc_key, {}).get("hbm_bytes_per_launch") if pmc_ok else None
        if layout_bytes is not None:
            # SURVEY 8d's storage model (1 B truth, 1 B code per cell) is the contract `frac` is quoted on; the kernels store truth as
            # 1 bit per cell and codes as 1 byte per 4-cell group, so what a launch MUST move is less: the honest yardstick for the
            # kernel's own efficiency and for the PMC traffic (traffic_over_layout_bytes > 1 = re-reads / partial lines)
            extra = dict(extra or {}, layout_bytes_per_launch=layout_bytes, frac_of_layout_bytes=layout_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         traffic_over_layout_bytes=(traffic / layout_bytes) if traffic else None)
        out = {"bound": "hbm", "kernel": timed["kernel"], "what": what, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": achieved / HBM_PEAK_GBS, "frac_raw": achieved / HBM_PEAK_GBS, "traffic": traffic,
               "traffic_source": f"{pmc_path} (static: separate rocprofv3 --pmc passes of this command, not measured in this run)"
               if traffic is not None else None,
               "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": us, "min_launch_us": timed["min_us"],
               "timed_launches": timed["launches"], "timing": TIMING,
               "stream_copy_GBps": copy_gbs, "frac_of_stream_copy": achieved / copy_gbs}
        out.update(extra or {})
        return out

    def fusion_bytes(c):   # 8 B per cell of the union (R+W once) + 1 B per (cell, message) code read
        return 8 * (c["fuse_local_cells"] + c["fuse_global_cells"]) + c["fuse_local_ops"] + c["fuse_global_ops"]

    k3 = times.get("sense")
    fuse = times.get("fuse")
    roofline = roofline_kernels = None
    if k3:
        # counters and durations of the roofline leg cover the same launches: its steps' K3 + its resets' start sensing
        cells = rl_counters["sense_cells"] / k3["launches"]
        roofline = roofline_entry("k_sense_update", "K3: sense + Bayes update of the footprint tiles", K3_BYTES_PER_CELL * cells, k3,
                                  {"algorithmic_bytes_per_cell": K3_BYTES_PER_CELL, "cells_per_launch": cells,
                                   "launch_envs": launch_envs,
                                   "launches": f"the K3 launches of {args.roofline_steps} steps ({rl_resets} resets among them) right after "
                                               "the timed region" + (f": the whole batch of {launch_envs} envs per launch on one stream, every launch alone "
                                               f"on the device (a second VecEnv of the same config); the timed region steps {args.streams} sub-batches of "
                                               f"{sub_envs} envs on {args.streams} streams, whose kernels run side by side: overlapped_us = their "
                                               "durations there, whole_step = what the device as a whole made of the HBM peak" if split else ""),
                                   "layout_bytes_per_cell": K3_LAYOUT_BYTES_PER_CELL},
                                  layout_bytes=K3_LAYOUT_BYTES_PER_CELL * cells)
        if overlapped:
            roofline["overlapped_us"] = {k: round(v["avg_us"], 2) for k, v in overlapped.items() if k in ("sense", "fuse", "plan", "reset_maps", "terrain")}
        # every algorithmic byte of a step of the TIMED region (K3 + fusion; the small plan kernel's ~2.4 MB left out) against
        # the step's wall time: what the whole step, launch gaps and resets included, makes of the HBM peak
        step_bytes = (K3_BYTES_PER_CELL * counters["sense_cells"] + fusion_bytes(counters)) / args.steps
        roofline["whole_step"] = {"algorithmic_bytes_per_step": step_bytes, "ms_per_step": 1e3 * dt / args.steps,
                                  "achieved": step_bytes / (dt / args.steps) / 1e9,
                                  "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS}
        # (kept under `roofline` as well: the driver's record keeps this object whole, top-level extras only by name)
        roofline["resets_timed"] = resets_timed
        roofline["steady_state"] = {"value": flying * ss_steps * world / ss_dt, "unit": "agent-env steps/s", "ms_per_step": 1e3 * ss_dt / ss_steps,
                                    "steps": ss_steps, "episodes": ss_steps // T, "resets": ss_resets,
                                    "whole_step_frac": (K3_BYTES_PER_CELL * ss_counters["sense_cells"] + fusion_bytes(ss_counters)) / ss_dt / 1e9 / HBM_PEAK_GBS,
                                    "note": "the timed loop continued for whole episodes (one reset per episode): the sustained rate"}
        roofline_kernels = [roofline]
    if fuse:
        by = fusion_bytes(rl_counters) / fuse["launches"]
        roofline_kernels.append(roofline_entry(
            "k_fuse_tiles" if "tiles" in fuse["kernel"] else "k_fuse_rows", "K4 local fusion + K5 global fusion and reward terms, one launch", by, fuse,
            {"algorithmic_bytes": "8 B per cell of the union (R+W once) + 1 B per (cell, message) code read",
             "local_cells_per_launch": rl_counters["fuse_local_cells"] / fuse["launches"],
             "global_cells_per_launch": rl_counters["fuse_global_cells"] / fuse["launches"],
             "message_cells_per_launch": (rl_counters["fuse_local_ops"] + rl_counters["fuse_global_ops"]) / fuse["launches"]},
            layout_bytes=(8 * (rl_counters["fuse_local_cells"] + rl_counters["fuse_global_cells"])
                          + 0.25 * (rl_counters["fuse_local_ops"] + rl_counters["fuse_global_ops"])) / fuse["launches"]))
    if roofline_kernels is not None:
        for cls, what in (("plan", "comm matrix + fusion plans + work list + K1, one wavefront per env"),
                          ("reset", "episode reset: device MT19937 scalars (and, with tracked area sums, the prior fills) per kernel launch"),
                          ("reset_maps", "episode reset: prior fill of the box each map was written in + start-position sensing, one launch"),
                          ("terrain", "random-field synthesis passes (per kernel launch)")):
            if cls in times:
                roofline_kernels.append({"kernel": times[cls]["kernel"], "what": what, "avg_launch_us": times[cls]["avg_us"],
                                         "min_launch_us": times[cls]["min_us"], "timed_launches": times[cls]["launches"],
                                         "us_per_step": times[cls]["avg_us"] * times[cls]["launches"] / args.roofline_steps})

    # Batch leg: the same shape at TWICE the metric's batch (config 2: 2048 envs).  Row-major rows get dearer per cell as a launch's maps grow, the
    # 128-byte tiles of ippm_set_map_layout do not, and map_layout="auto" takes them from 2 GB of maps on -- so the chip has more to give than the
    # metric's 1024 envs show.  NOT the metric's configuration: reported beside it under roofline.batch_leg, never as `value`.
    batch_leg = None
    if args.batch_leg and world == 1 and not args.graphs and not teams:
        try:
            env = roof_env = first = None
            torch.cuda.empty_cache()
            big_E = 2 * args.envs
            big = SplitVecEnv(params, big_E, parts=max(args.streams, 1), device=device, philox_seed=3, terrain=args.terrain) if split else \
                VecEnv(params, big_E, device=device, philox_seed=3, terrain=args.terrain, track_area=False)
            if args.placement_draws > 1:
                for part in (big.parts if split else [big]):
                    part.tune_placement(args.placement_draws)
            bwave = [0]
            if split:
                big.start(lambda w: episode_ids(1, w, big_E, rank, world))
                bstep = lambda: big.advance(POLICY_UNIFORM)   # noqa: E731
            else:
                bt = {"t": 0}
                big.reset(episode_ids(1, 0, big_E, rank, world))

                def bstep():
                    big.steps(bt["t"], policy=POLICY_UNIFORM, features=False)
                    bt["t"] += 1
                    if bt["t"] == T:
                        bwave[0] += 1
                        big.reset(episode_ids(1, bwave[0], big_E, rank, world))
                        bt["t"] = 0
            for _ in range(T):
                bstep()
            torch.cuda.synchronize()
            b0 = time.perf_counter()
            b_steps = 10 * T
            for _ in range(b_steps):
                bstep()
            torch.cuda.synchronize()
            b_dt = time.perf_counter() - b0
            batch_leg = {"envs_per_gpu": big_E, "value": big_E * N * b_steps / b_dt, "unit": "agent-env steps/s", "ms_per_step": 1e3 * b_dt / b_steps,
                         "steps": b_steps, "episodes": 10, "resets": 10, "faults": int(big.fault.abs().sum()),
                         "map_layout": "tiles" if big.tiled else "rows",
                         "note": f"the same shape and loop at {big_E} envs (twice the metric's batch): 10 whole episodes, one reset per episode; NOT the metric's configuration"}
            big = None
            torch.cuda.empty_cache()
        except Exception as exc:   # noqa: BLE001  (an extra leg must never cost the line)
            batch_leg = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    if batch_leg is not None and isinstance(roofline, dict):
        roofline["batch_leg"] = batch_leg
    coma = collective = None
    if args.train_rounds > 0:
        # BASELINE configs[2]: full COMA actor + counterfactual critic training on the same env config
        from ippmarl.trainer import COMATrainer
        env = roof_env = first = None  # release the env-only state before the trainer allocates its own
        torch.cuda.empty_cache()
        tr = COMATrainer(params, args.envs, device=device, philox_seed=3, rank=rank, world=world, terrain=args.terrain,
                         placement_draws=args.placement_draws, team_sizes=teams)
        tr.rollout("train")
        tr.update()  # warm-up round (MIOpen kernel selection, allocator)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        c0 = time.perf_counter()
        roll_s = 0.0
        for _ in range(args.train_rounds):
            r0 = time.perf_counter()
            tr.rollout("train")
            torch.cuda.synchronize()
            roll_s += time.perf_counter() - r0
            stats = tr.update()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        cdt = time.perf_counter() - c0
        coma = {"updates_per_s": args.train_rounds / cdt, "s_per_update": cdt / args.train_rounds,
                "transitions_per_update": stats["transitions"], "adam_steps_per_update": stats["adam_steps"],
                "rollout_agent_env_steps_per_s": flying * tr.T * world * args.train_rounds / roll_s,
                "note": "one update = TD(lambda) targets + data_passes x batch_number minibatch steps of critic and actor "
                        "(reference round: 25+25 Adam steps on 300 transitions); nets float32 in PyTorch-ROCm"}
        # kernel times of one more learned-policy rollout (area sums tracked by K3/K4/K5, K6 = 121-cell assembly), untimed above.
        # A training-mode rollout (actions sampled from the epsilon-mixed policy): a greedy one after two updates of a random net
        # sends every UAV wherever that net happens to point -- all down to 5 m in one process, all up in the next -- and the map
        # kernels' work with them (sense 27 .. 49 us, fuse 55 .. 116 us between processes); the cells per step say what was done.
        tr.env.profile = True
        tr.env.counters(reset=True)
        tr.rollout("train")
        tr.env.profile = False
        kt = tr.env.event_times_us()
        cc = tr.env.counters()
        coma["rollout_kernel_us"] = {k: round(v["avg_us"], 2) for k, v in kt.items()}
        coma["rollout_kernel_us"]["cells_per_step"] = {"sense": cc["sense_cells"] / tr.T, "fuse_local": cc["fuse_local_cells"] / tr.T,
                                                       "fuse_global": cc["fuse_global_cells"] / tr.T}
        coma["rollout_kernel_us"]["kernels"] = {k: v["kernel"] for k, v in kt.items()}
        coma["rollout_kernel_us"]["note"] = ("dispatch-bound start/stop events (kernel-only durations); sense/fuse here also maintain "
                                             "the 11x11 area sums of every map, which is what lets the K6 builders skip the maps")
        if roofline_kernels is not None and "actor_features" in kt:
            # K6 no longer reads the maps (SURVEY 8d priced it at 4 B per cell of the N+1 maps per step): its inputs, the 11 x 11 area
            # sums, are kept current by the kernels that write maps.  What network inputs cost is therefore K6's own launches PLUS what
            # tracking adds to K3 and to the fusion -- the difference between the rollout's tracked kernels and the env-only step's
            # untracked ones (different cells per step: a learned policy flies elsewhere than a random one; both are given).
            k6_us = sum(kt[k]["avg_us"] for k in ("actor_features", "critic_features"))
            untracked = {"sense": times.get("sense", {}).get("avg_us"), "fuse": times.get("fuse", {}).get("avg_us")}
            roofline_kernels.append({
                "kernel": "K6 (k_actor_features + k_critic_features) on tracked area sums", "what": "network inputs: the feature builders' own "
                "launches plus what keeping the area sums current adds to K3 and to the fusion (no roofline fraction: the maps are not read)",
                "k6_us": k6_us,
                "tracking_cost_us": {k: (kt[k]["avg_us"] - untracked[k]) if k in kt and untracked[k] else None for k in ("sense", "fuse")},
                "tracked_us": {k: kt[k]["avg_us"] for k in ("sense", "fuse") if k in kt},
                "untracked_us": untracked,
                "cells_per_step": {"tracked_rollout": coma["rollout_kernel_us"]["cells_per_step"],
                                   "untracked_env_only": ({"sense": rl_counters["sense_cells"] / args.roofline_steps,
                                                           "fuse_local": rl_counters["fuse_local_cells"] / args.roofline_steps,
                                                           "fuse_global": rl_counters["fuse_global_cells"] / args.roofline_steps}
                                                          if rl_counters else None)},
                "streaming_form_bytes_per_step": 4.0 * (N + 1) * grid[0] * grid[1] * E})
        if world > 1:   # evidence that the gradient exchange really ran over `world` ranks: every rank's device, calls and bytes
            import socket
            mine = {"rank": rank, "host": socket.gethostname(), "device": torch.cuda.current_device(),
                    "name": torch.cuda.get_device_name(), "allreduce_calls": tr.reducer.calls,
                    "allreduce_bytes": tr.reducer.bytes_reduced}
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            if dist.get_backend() == "nccl" and len({(r["host"], r["device"]) for r in gathered}) != world:
                raise SystemExit(f"{world} ranks over RCCL but only {len({(r['host'], r['device']) for r in gathered})} distinct (host, device) "
                                 f"pairs: {gathered}")
            collective = {"backend": dist.get_backend(), "library": "RCCL over xGMI" if dist.get_backend() == "nccl" else dist.get_backend(),
                          "world_size": dist.get_world_size(), "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset"),
                          "launcher_ipc_setting": os.environ.get("IPPM_BENCH_IPC_SETTING"), "allreduce_calls": tr.reducer.calls,
                          "allreduce_bytes": tr.reducer.bytes_reduced,
                          "per_update": {"calls": tr.reducer.calls // (args.train_rounds + 1),
                                         "bytes": tr.reducer.bytes_reduced // (args.train_rounds + 1)},
                          "ranks": gathered}
        if world == 1 and not teams:
            # the reference's own round size for comparison with its 0.164 updates/s: 5 episodes ->
            # 300 transitions per update, 25 + 25 Adam steps on 60-sample minibatches
            del tr
            torch.cuda.empty_cache()
            per_episode = T * N
            ref_envs = max(1, -(-params["networks"]["batch_size"] * params["networks"]["batch_number"] // per_episode))
            tr = COMATrainer(params, ref_envs, device=device, philox_seed=3, terrain=args.terrain, graphs=True)
            tr.rollout("train")
            tr.update()
            torch.cuda.synchronize()

            def timed_rounds(rounds):
                c0 = time.perf_counter()
                for _ in range(rounds):
                    tr.rollout("train")
                    st = tr.update()
                torch.cuda.synchronize()
                return rounds / (time.perf_counter() - c0), st

            eager_rate, _ = timed_rounds(5)
            # this round is ~3000 launches of a few microseconds each: recorded into hipGraphs (16 rollout-step graphs + one graph
            # of the whole update: TD targets and the 25 + 25 Adam steps), it runs at the kernels' pace instead of Python's
            tr.capture_graphs()
            timed_rounds(2)
            rate, stats = timed_rounds(20)
            coma["reference_sized_round"] = {"updates_per_s": rate, "updates_per_s_eager": eager_rate,
                                             "transitions_per_update": stats["transitions"], "envs": ref_envs,
                                             "adam_steps_per_update": stats["adam_steps"],
                                             "hip_graphs": "16 rollout-step graphs + 1 update graph per round (COMATrainer.capture_graphs)"}
    seam = None
    if rank == 0 and not args.no_dropin_seam and not teams:
        env = roof_env = first = None
        torch.cuda.empty_cache()
        seam = dropin_seam(device)
    # per-rank audit trail: each rank's own clock around the timed region and how its placement search ended, so that a multi-GPU
    # line can be checked rank by rank (value = the units all ranks processed / the slowest rank's time)
    free_b, total_b = torch.cuda.mem_get_info()
    mine = {"rank": rank, "device": torch.cuda.current_device(), "peak_allocated_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 3),
            "device_memory_in_use_GB": round((total_b - free_b) / 2 ** 30, 2), "placement_note": placement_note,
            "ms_per_step": 1e3 * dt_rank / args.steps,
            "agent_env_steps_per_s": flying * args.steps / dt_rank,
            "placement_stopped": [(p or {}).get("stopped") for p in (placement if isinstance(placement, list) else [placement])],
            "placement_draws": [(p or {}).get("draws") for p in (placement if isinstance(placement, list) else [placement])]}
    per_rank = [mine]
    if dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        total_steps = flying * args.steps * world
        is_c1 = (N, grid[0], E) == (4, 256, 1024) and env_actions == 6 and not args.episode_comm_range and not teams
        shape = ((f"teams of {args.team_sizes} of " if teams else "") + f"{N} UAVs, {grid[0]}x{grid[1]} grid, {E} batched envs per GPU, random policy over {env_actions} actions"
                 f"{', per-episode comm range' if args.episode_comm_range else ''}, env-step kernels only")
        out = {
            "metric": f"agent-env steps/s ({N} UAVs, {grid[0]}x{grid[1]} grid, random policy, env-step HIP kernels)",
            "value": total_steps / dt, "unit": "agent-env steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": ("synthetic random-field terrain (power-law Gaussian random field thresholded at 0.5, generated on the device per "
                     "episode; Philox sensor noise)" if args.terrain == "random_field" else
                     "synthetic (device-generated half-plane truth, Philox sensor noise)"),
            "config": {"workload": ("BASELINE.json configs[1]: " if is_c1 else "NOT the metric's config (a parity-test shape): ") + shape,
                       "envs_per_gpu": E, "n_agents": N, "grid": grid,
                       "episode_steps": T, "terrain": args.terrain, "parallelism": f"env-sharded x{world} (no data-path collective)",
                       "streams": args.streams if split else 1, "staggered_episodes": bool(split and args.stagger), "envs_per_launch": sub_envs,
                       "launches_per_step": 3 * (args.streams if split else 1), "stream_check": stream_check, "hip_graphs": bool(args.graphs),
                       "roofline_steps": args.roofline_steps,
                       "map_layout": map_layout,
                       "terrain_prefetch": bool(args.terrain_prefetch and args.terrain == "random_field")},
            "ranks": world,
            "resets_timed": resets_timed,
            "steady_state": {"ms_per_step": 1e3 * ss_dt / ss_steps, "value": flying * ss_steps * world / ss_dt, "steps": ss_steps, "resets": ss_resets,
                             "note": f"the same loop continued for {ss_steps} steps = {ss_steps // T} whole episodes (exactly one reset per {T} steps, as in an "
                                     f"endless run); `value` above is the driver's window of --steps {args.steps}, which held {resets_timed} reset(s), "
                                     f"i.e. one per {args.steps / max(resets_timed, 1):.1f} steps" + ("" if resets_timed else " (none at all)")
                                     + ": quote steady_state for sustained throughput"},
            "per_rank": {"ranks": per_rank, "slowest_rank": max(per_rank, key=lambda r: r["ms_per_step"])["rank"],
                         "ms_per_step_min": min(r["ms_per_step"] for r in per_rank), "ms_per_step_max": max(r["ms_per_step"] for r in per_rank),
                         "value_sum_of_ranks": sum(r["agent_env_steps_per_s"] for r in per_rank), "value_from_max_time": total_steps / dt},
            "collective": collective if args.train_rounds > 0 else None,
            "faults": faults,
            "cells": counters,
            "roofline": roofline,
            "roofline_kernels": roofline_kernels,
            "placement": placement, "roofline_leg_placement": roof_placement,
            "coma_training": coma,
            "dropin_seam": seam,
        }
    if dist:   # every collective is done: the other ranks may leave while rank 0 times the CPU baseline on the host cores
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
