#!/usr/bin/env python
"""bench.py -- agent-env steps/s of the batched multi-UAV environment step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 150 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 150 --warmup 30

A "step" is one environment step of ALL envs on a rank, three launches: comm matrix + fusion plans + mask/act/move with a
uniform random valid policy (K1) -> local fusion (K4) and global fusion + reward terms (K5) -> sense + Bayes update at the
new positions (K3, which also completes the reward); every 15 steps the envs are reset to fresh episodes inside the timed
region (device-side MT19937 + Philox, terrain synthesis included).
Default workload = BASELINE.json configs[1]: 4 UAVs, 256 x 256 grid, 1024 batched envs per GPU, random policy.
Inputs are synthetic and resident in HBM (truth fields generated on the device).  Envs are independent, so N GPUs
shard envs with no data-path collective ("weak" scaling: 1024 envs per GPU).

The JSON line also carries
  roofline         : the map-update kernel K3 (sense_update): algorithmic bytes = 10 B per footprint cell (4R+4W posterior,
                     1R truth, 1W measurement code) / HIP-event time of every K3 launch of the timed region
  roofline_kernels : the same for the fusion launch (K4+K5: 8 B per cell of the union + 1 B per (cell, message)) and, when
                     training is on, the K6 feature builders
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
# the unmodified reference, measured in the build container on 8 host cores; it cannot travel to the GPU box
REFERENCE_PROBE = {"agent_env_steps_per_s": 10.0, "coma_updates_per_s": 0.164, "cores": 8,
                   "source": "SURVEY.md section 6: /root/reference run in the build container (default params, 493x493)"}
PIXELS = {128: 15, 256: 30, 512: 60, 1024: 120}


def bench_params(args):
    from ippmarl.params import grid256_params
    number = PIXELS[args.grid]   # other BASELINE.json grid sizes are parity-test configs; 256 is the metric's
    return grid256_params(experiment__missions__n_agents=args.agents, sensor__pixel__number_x=number, sensor__pixel__number_y=number)


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
    return {"value": steps1 / dt1, "unit": "agent-env steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps1} agent-env steps of 1 env ({shape}) in {dt1:.1f}s of the NumPy oracle on 1 of {cores} host cores",
            "many_env": {"value": stepsN / dtN, "unit": "agent-env steps/s", "cores": procs, "envs": per * procs,
                         "sample": f"{stepsN} agent-env steps of {per * procs} envs in {dtN:.1f}s, {procs} oracle processes "
                                   f"({per} envs each, one core each)"},
            "reference_probe": REFERENCE_PROBE}


def newest_pmc_summary():
    """profiles/rNN/pmc_summary.json of the newest round that has one (written by tools/pmc_summary.py from separate
    rocprofv3 --pmc passes of this command)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_summary.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--grid", type=int, default=256, choices=sorted(PIXELS))
    ap.add_argument("--terrain", default="random_field", choices=["random_field", "split"],
                    help="ground truth: the power-law random field of ground_truths.py:25-40 generated on the device, or "
                         "the half-plane split the reference flies over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-events", type=int, default=4, help="bracket the K3 and fusion launches of every K-th step (and "
                    "every K-th reset) of the timed region with HIP events for the roofline legs; an event pair costs ~3 us of "
                    "stream time, so K=1 adds ~13 us to every step; 0 = no brackets")
    ap.add_argument("--train-rounds", type=int, default=2, help="COMA rounds (rollout with the actor + full update) timed after "
                    "the env-only region for the COMA updates/s figure; 0 disables")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo lets the "
                    "multi-rank code path be exercised on a single GPU)")
    ap.add_argument("--graphs", type=int, default=0, help="replay {plan+K1, K4+K5} of each step from a hipGraph; K3 stays an "
                    "ordinary launch.  Off by default: three launches per step do not need it")
    ap.add_argument("--calib", action="store_true", help="PMC calibration: 3 device-to-device copies of the local maps (known "
                    "bytes read and written by a 16 B/lane streaming kernel) before the timed loop")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n_dev = torch.cuda.device_count()
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank % n_dev}"))
        else:
            dist.init_process_group(args.dist_backend)
        local_rank %= n_dev
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(device)

    from ippmarl import _ffi
    from ippmarl.parallel import episode_ids
    from ippmarl.vec_env import VecEnv, POLICY_UNIFORM
    params = bench_params(args)
    # env-only stepping never builds network inputs: the area sums are not tracked here (the trainer below tracks them)
    env = VecEnv(params, args.envs, device=device, philox_seed=3, terrain=args.terrain, track_area=False)
    E, N, T = env.E, env.d.n_agents, env.d.budget + 1
    wave = [0]

    def reset():
        env.reset(episode_ids(1, wave[0], E, rank, world))   # disjoint episodes per rank and wave
        wave[0] += 1

    sample = [0]

    def sampled():   # every K-th launch group of the timed region carries event brackets
        sample[0] += 1
        return bool(args.profile_events) and sample[0] % args.profile_events == 0

    def one_step(t, timed):
        env.profile = timed and sampled()
        if args.graphs:
            env.step_graphed(t)
        else:
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.profile = False

    reset()
    if args.graphs:
        env.capture_step_graphs(POLICY_UNIFORM)
    scratch = torch.empty_like(env.local)

    def stream_copy():
        env.ctx.call("ippm_stream_copy", env._p(env.local), _ffi.ptr(scratch), env.local.numel() * 4, env.stream)

    if args.calib:
        for _ in range(3):
            stream_copy()
        torch.cuda.synchronize()
    t_in_ep = 0
    for _ in range(args.warmup):
        one_step(t_in_ep, False)
        t_in_ep += 1
        if t_in_ep == T:
            reset()
            t_in_ep = 0
    env.counters(reset=True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    resets_timed = 0
    for _ in range(args.steps):
        one_step(t_in_ep, True)
        t_in_ep += 1
        if t_in_ep == T:
            resets_timed += 1
            env.profile = sampled()   # the reset's start-position sensing is a K3 launch of the timed region too
            reset()
            env.profile = False
            t_in_ep = 0
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    counters = env.counters()
    times = env.event_times_us()
    faults = int(env.fault.abs().sum())
    grid = [env.d.grid_x, env.d.grid_y]

    # cost of an empty event bracket on this stream (the bracketed times include it; rocprofv3's kernel trace does not)
    empty = []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        empty.append((a, b))
    torch.cuda.synchronize()
    event_overhead_us = 1e3 * sum(a.elapsed_time(b) for a, b in empty) / len(empty)
    # second denominator: what a plain 16 B/lane device-to-device copy of the local maps reaches on this box
    stream_copy()
    ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ca.record()
    for _ in range(3):
        stream_copy()
    cb.record()
    torch.cuda.synchronize()
    copy_gbs = 3 * 2 * env.local.numel() * 4 / (ca.elapsed_time(cb) * 1e-3) / 1e9
    del scratch
    pmc, pmc_path = newest_pmc_summary()
    pmc_ok = bool(pmc) and pmc.get("envs_per_gpu") == E and pmc.get("n_agents") == N and pmc.get("grid") == grid[0]

    def roofline_entry(kernel, pmc_key, label, bytes_per_launch, bracket, extra=None):
        if bracket is None or not bracket["launches"]:
            return None
        raw_us = bracket["avg_us"]
        # kernel duration = bracketed time minus the cost of the bracket itself, calibrated above on the same stream; the
        # uncorrected figures are kept as *_raw (rocprofv3's kernel-only duration of the same command: profiles/rNN/)
        us = max(raw_us - event_overhead_us, 0.5 * raw_us)
        achieved = bytes_per_launch / (us * 1e-6) / 1e9
        traffic = pmc.get(pmc_key, {}).get("hbm_bytes_per_launch") if pmc_ok else None
        out = {"bound": "hbm", "kernel": label, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
               "traffic_source": f"{pmc_path} (static: separate rocprofv3 --pmc passes of this command, not measured in this run)"
               if traffic is not None else None,
               "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": us, "avg_launch_us_raw": raw_us,
               "bracketed_launches": bracket["launches"], "empty_event_pair_us": event_overhead_us,
               "frac_raw": bytes_per_launch / (raw_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
               "stream_copy_GBps": copy_gbs, "frac_of_stream_copy": achieved / copy_gbs}
        out.update(extra or {})
        return out

    k3 = times.get("sense")
    fuse = times.get("fuse")
    roofline = roofline_kernels = None
    if k3:
        # the counters cover every launch of the timed region, the brackets a 1-in-K sample of them
        k3_launches, fuse_launches = args.steps + resets_timed, args.steps
        cells = counters["sense_cells"] / k3_launches
        roofline = roofline_entry("k_sense_update", "k_sense_update", "k_sense_update (K3: sense + Bayes update of the footprint tile)",
                                  K3_BYTES_PER_CELL * cells, k3,
                                  {"algorithmic_bytes_per_cell": K3_BYTES_PER_CELL, "cells_per_launch": cells,
                                   "note": "avg_launch_us = HIP-event brackets of the K3 launches of every K-th step of the timed region "
                                           "(config.event_brackets_every) minus the cost of an "
                                           "empty event pair measured in the same run; frac_raw keeps the uncorrected figure"})
        roofline_kernels = [roofline]
    if fuse:
        lc, lo = counters["fuse_local_cells"], counters["fuse_local_ops"]
        gc, go = counters["fuse_global_cells"], counters["fuse_global_ops"]
        by = (8 * (lc + gc) + (lo + go)) / fuse_launches
        roofline_kernels.append(roofline_entry(
            "k_fuse_rows", "k_fuse_rows", "k_fuse_rows (K4 local fusion + K5 global fusion and reward terms, one launch)", by, fuse,
            {"algorithmic_bytes": "8 B per cell of the union (R+W once) + 1 B per (cell, message) code read",
             "local_cells_per_launch": lc / fuse_launches, "global_cells_per_launch": gc / fuse_launches,
             "message_cells_per_launch": (lo + go) / fuse_launches}))

    coma = None
    if args.train_rounds > 0:
        # BASELINE configs[2]: full COMA actor + counterfactual critic training on the same env config
        from ippmarl.trainer import COMATrainer
        env = None  # release the env-only state before the trainer allocates its own
        torch.cuda.empty_cache()
        tr = COMATrainer(params, args.envs, device=device, philox_seed=3, rank=rank, world=world, terrain=args.terrain)
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
                "rollout_agent_env_steps_per_s": tr.E * tr.N * tr.T * world * args.train_rounds / roll_s,
                "note": "one update = TD(lambda) targets + data_passes x batch_number minibatch steps of critic and actor "
                        "(reference round: 25+25 Adam steps on 300 transitions); nets float32 in PyTorch-ROCm"}
        # kernel times of one more learned-policy rollout (area sums tracked by K3/K4/K5, K6 = 121-cell assembly), untimed above
        tr.env.profile = True
        tr.env.counters(reset=True)
        tr.rollout("eval")
        tr.env.profile = False
        kt = tr.env.event_times_us()
        coma["rollout_kernel_us"] = {k: round(max(v["avg_us"] - event_overhead_us, 0.5 * v["avg_us"]), 2) for k, v in kt.items()}
        coma["rollout_kernel_us"]["note"] = ("HIP-event brackets minus the empty-bracket cost; sense/fuse here also maintain the "
                                             "11x11 area sums of every map, which is what lets the K6 builders skip the maps")
        if roofline_kernels is not None and "actor_features" in kt:
            c = tr.env.counters()
            k6_us = sum(max(kt[k]["avg_us"] - event_overhead_us, 0.5 * kt[k]["avg_us"]) for k in ("actor_features", "critic_features"))
            roofline_kernels.append({
                "bound": "hbm", "kernel": "K6 (k_actor_features + k_critic_features) with tracked area sums", "unit": "GB/s",
                "peak": HBM_PEAK_GBS, "avg_launch_us": k6_us,
                "algorithmic_bytes_per_launch": 4.0 * (N + 1) * grid[0] * grid[1] * E,
                "achieved": 4.0 * (N + 1) * grid[0] * grid[1] * E / (k6_us * 1e-6) / 1e9,
                "frac": 4.0 * (N + 1) * grid[0] * grid[1] * E / (k6_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "traffic": None,
                "note": "SURVEY 8d prices K6 at 4 B per cell of the N+1 maps per env step (one streaming read); the maps are no "
                        "longer read at all (their area sums are maintained by the kernels that write them), so the 'achieved' "
                        "figure is the rate a streaming implementation would need to match this time and may exceed the peak",
                "sense_cells_per_step": c["sense_cells"] / tr.T})
        if world == 1:
            # the reference's own round size for comparison with its 0.164 updates/s: 5 episodes ->
            # 300 transitions per update, 25 + 25 Adam steps on 60-sample minibatches
            del tr
            torch.cuda.empty_cache()
            per_episode = T * N
            ref_envs = max(1, -(-params["networks"]["batch_size"] * params["networks"]["batch_number"] // per_episode))
            tr = COMATrainer(params, ref_envs, device=device, philox_seed=3, terrain=args.terrain)
            tr.rollout("train")
            tr.update()
            torch.cuda.synchronize()
            rounds = 5
            c0 = time.perf_counter()
            for _ in range(rounds):
                tr.rollout("train")
                stats = tr.update()
            torch.cuda.synchronize()
            cdt = time.perf_counter() - c0
            coma["reference_sized_round"] = {"updates_per_s": rounds / cdt, "transitions_per_update": stats["transitions"],
                                             "envs": ref_envs, "adam_steps_per_update": stats["adam_steps"]}
    if rank == 0:
        total_steps = E * N * args.steps * world
        is_c1 = (N, grid[0], E) == (4, 256, 1024)
        shape = f"{N} UAVs, {grid[0]}x{grid[1]} grid, {E} batched envs per GPU, random policy, env-step kernels only"
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
                       "launches_per_step": 3, "hip_graphs": bool(args.graphs), "event_brackets_every": args.profile_events},
            "faults": faults,
            "cells": counters,
            "roofline": roofline,
            "roofline_kernels": roofline_kernels,
            "coma_training": coma,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
