#!/usr/bin/env python
"""bench.py -- agent-env steps/s of the batched multi-UAV environment step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 150 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 150 --warmup 30

A "step" is one environment step of ALL envs on a rank: comm matrix + local fusion (K4), global fusion + reward
(K5), mask/act/move with a uniform random valid policy (K1), sense + Bayes update at the new positions (K3);
every 15 steps the envs are reset to fresh episodes inside the timed region (device-side MT19937 + Philox).
Workload = BASELINE.json configs[1]: 4 UAVs, 256 x 256 grid, 1024 batched envs per GPU, random policy.
Inputs are synthetic and resident in HBM (truth fields generated on the device).  Envs are independent, so N GPUs
shard envs with no data-path collective ("weak" scaling: 1024 envs per GPU).

The JSON line also carries
  roofline     : dominant kernel = K3 sense_update; algorithmic bytes = 10 B per footprint cell (4R+4W posterior,
                 1R truth, 1W measurement code) / HIP-event time of the K3 launches in the timed region
  cpu_baseline : the NumPy oracle (a port of the reference's CPU path, parity-pinned against it) stepping the same
                 config one env at a time on one host core, bounded to ~15 s
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
K3_BYTES_PER_CELL = 10


def bench_params(args):
    from ippmarl.params import grid256_params
    if args.grid == 256:
        return grid256_params(experiment__missions__n_agents=args.agents)
    number = {128: 15, 512: 60, 1024: 120}[args.grid]   # other BASELINE.json grid sizes (parity-test configs; not the metric)
    return grid256_params(experiment__missions__n_agents=args.agents, sensor__pixel__number_x=number, sensor__pixel__number_y=number)


def cpu_baseline(params, budget_s=15.0, terrain="random_field"):
    """Oracle (kind 'port'): same env-only workload, one env at a time, explicit NumPy on one core."""
    import ipp_oracle as O
    torch.set_num_threads(1)
    d = O.Derived(params)
    seed = 3
    steps, t0, episode = 0, time.perf_counter(), 1
    while time.perf_counter() - t0 < budget_s:
        holder = {}

        def correctness(i, s, shape):
            pos = holder["ep"].agents[i]["position"]
            _, fc = O.project_field_of_view(d, pos)
            return O.philox_correctness(seed, episode, i, s, fc, d.gy, O.noise_of_altitude(pos[2]))

        ep = O.OracleEpisode(params, episode, correctness,
                             lambda i, t, m, o: O.uniform_valid_action(O.philox_action_word(seed, episode, i, t), m),
                             build_features=False,
                             truth=O.grf_field(d.gx, d.gy, episode, float(params["sensor"]["simulation"]["cluster_radius"]))
                             if terrain == "random_field" else None)
        holder["ep"] = ep
        for t in range(d.budget + 1):
            ep.step(t)
            steps += d.n_agents
            if time.perf_counter() - t0 > budget_s:
                break
        episode += 1
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "agent-env steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} agent-env steps ({episode - 1} episodes of 1 env, 4 UAVs, 256x256, env-only) in {dt:.1f}s "
                      f"of NumPy oracle on 1 of {os.cpu_count()} host cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--grid", type=int, default=256, choices=[128, 256, 512, 1024])
    ap.add_argument("--terrain", default="random_field", choices=["random_field", "split"],
                    help="ground truth: the power-law random field of ground_truths.py:25-40 generated on the device, or "
                         "the half-plane split the reference flies over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-events", type=int, default=1, help="time every K3 launch with HIP events (roofline)")
    ap.add_argument("--train-rounds", type=int, default=2, help="COMA rounds (rollout with the actor + full update) timed after "
                    "the env-only region for the COMA updates/s figure; 0 disables")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo lets the "
                    "multi-rank code path be exercised on a single GPU)")
    ap.add_argument("--graphs", type=int, default=1, help="replay the launch-bound part of each step (comm, K4, K5, K1) from "
                    "hipGraphs; K3 stays an ordinary launch bracketed by events")
    ap.add_argument("--calib", action="store_true", help="PMC calibration: 3 device-to-device clones of the local maps (known "
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

    from ippmarl.vec_env import VecEnv, POLICY_UNIFORM
    params = bench_params(args)
    env = VecEnv(params, args.envs, device=device, philox_seed=3, terrain=args.terrain)
    E, N, T = env.E, env.d.n_agents, env.d.budget + 1
    base = torch.arange(1, E + 1, dtype=torch.int64) + rank * E  # disjoint episodes per rank
    wave = [0]
    ev_pairs = []

    def reset():
        env.reset(base + wave[0] * E * world)
        wave[0] += 1

    timing = [False]
    raw_sense = env.sense

    def timed_sense(stage, flips=None, agent=-1):
        """Every K3 launch of the timed region (step sensing and reset sensing) is bracketed by HIP events on the
        stream it is launched on (torch's current stream)."""
        if timing[0] and args.profile_events:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            raw_sense(stage, flips, agent)
            b.record()
            ev_pairs.append((a, b))
        else:
            raw_sense(stage, flips, agent)

    env.sense = timed_sense

    def one_step(t, timed):
        timing[0] = timed
        if args.graphs:
            env.step_graphed(t)
        else:
            env.build_observations(t, features=False)
            env.steps(t, policy=POLICY_UNIFORM, features=False)

    reset()
    if args.graphs:
        env.capture_step_graphs(POLICY_UNIFORM)
    if args.calib:
        for _ in range(3):
            env.local.clone()
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
    for _ in range(args.steps):
        one_step(t_in_ep, True)
        t_in_ep += 1
        if t_in_ep == T:
            reset()
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
    faults = int(env.fault.abs().sum())
    grid = [env.d.grid_x, env.d.grid_y]

    k3_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) if ev_pairs else None
    # cost of an empty event bracket on this stream (the bracketed K3 time above includes it; rocprofv3's kernel trace does not)
    empty = []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        empty.append((a, b))
    torch.cuda.synchronize()
    event_overhead_us = 1e3 * sum(a.elapsed_time(b) for a, b in empty) / len(empty)
    sense_cells_step = counters["sense_cells"]
    # second denominator: what a plain device-to-device copy of the local maps reaches on this box
    src = env.local
    dst = torch.empty_like(src)
    dst.copy_(src)
    ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ca.record()
    for _ in range(3):
        dst.copy_(src)
    cb.record()
    torch.cuda.synchronize()
    copy_gbs = 3 * 2 * src.numel() * 4 / (ca.elapsed_time(cb) * 1e-3) / 1e9
    del dst
    roofline = None
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.isfile(pmc_file):  # produced by tools/pmc_summary.py from separate rocprofv3 --pmc passes of this command
        with open(pmc_file) as f:
            pmc = json.load(f)
        if pmc.get("envs_per_gpu") == E and pmc.get("n_agents") == N:
            traffic = pmc.get("k_sense_update", {}).get("hbm_bytes_per_launch")
    if k3_ms:
        # kernel duration = bracketed time minus the cost of the bracket itself, calibrated just above on the same stream
        # (an empty pair reads ~5 us; rocprofv3's kernel trace of the same command agrees with the corrected figure)
        raw_us = 1e3 * k3_ms / len(ev_pairs)
        launch_us = max(raw_us - event_overhead_us, 0.5 * raw_us)
        achieved = K3_BYTES_PER_CELL * sense_cells_step / len(ev_pairs) / (launch_us * 1e-6) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_sense_update (K3: sense + Bayes update of the footprint tile)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "algorithmic_bytes_per_launch": K3_BYTES_PER_CELL * sense_cells_step / max(len(ev_pairs), 1),
                    "algorithmic_bytes_per_cell": K3_BYTES_PER_CELL,
                    "cells_per_launch": sense_cells_step / max(len(ev_pairs), 1),
                    "avg_launch_us": launch_us, "avg_launch_us_raw": raw_us, "launches": len(ev_pairs),
                    "empty_event_pair_us": event_overhead_us,
                    "stream_copy_GBps": copy_gbs, "frac_of_stream_copy": achieved / copy_gbs,
                    "frac_raw": K3_BYTES_PER_CELL * sense_cells_step / len(ev_pairs) / (raw_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "note": "avg_launch_us = event-bracketed time of every K3 launch of the timed region minus the cost of an "
                            "empty event pair measured in the same run (avg_launch_us_raw / frac_raw keep the uncorrected "
                            "figures); profiles/r01/kernel_stats_*.csv is rocprofv3's kernel-only duration of the same command"}

    coma = None
    if args.train_rounds > 0:
        # BASELINE configs[2]: full COMA actor + counterfactual critic training on the same env config
        from ippmarl.trainer import COMATrainer
        env.sense = None
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
        out = {
            "metric": "agent-env steps/s (4 UAVs, 256x256 grid, random policy, env-step HIP kernels)",
            "value": total_steps / dt, "unit": "agent-env steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": ("synthetic random-field terrain (power-law Gaussian random field thresholded at 0.5, generated on the device per "
                     "episode; Philox sensor noise)" if args.terrain == "random_field" else
                     "synthetic (device-generated half-plane truth, Philox sensor noise)"),
            "config": {"workload": "BASELINE.json configs[1]: 4 UAVs, 256x256 grid, 1024 batched envs per GPU, random policy, "
                                   "env-step kernels only", "envs_per_gpu": E, "n_agents": N, "grid": grid,
                       "episode_steps": T, "terrain": args.terrain, "parallelism": f"env-sharded x{world} (no data-path collective)", "hip_graphs": bool(args.graphs)},
            "faults": faults,
            "cells": counters,
            "roofline": roofline,
            "coma_training": coma,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(params, terrain=args.terrain)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
