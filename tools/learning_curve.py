#!/usr/bin/env python
"""Does the COMA loop learn?  Trains the actor / counterfactual critic on BASELINE config 3 (4 UAVs, 256 x 256, the reference's
params.yaml hyper-parameters) and scores the GREEDY policy every few updates on a FIXED set of evaluation episodes, next to the
uniform random walk and the greedy information-gain planner on the same episodes (same truth, start cells and sensor noise: every
random stream is keyed by the episode number).  The reference's own yardstick: coma_test.py:84-97,177-196 (greedy deployment),
random_baseline.py:91-96, IG_baseline.py:127-148; training cadence missions/coma_mission.py:48-172.

    python tools/learning_curve.py --envs 1024 --updates 60 --eval-every 10 --out profiles/r06/learning_curve_1024.json
    python tools/learning_curve.py --envs 5 --graphs --updates 2000 --eval-every 100 --eval-envs 256 ...   # the reference's round size

One update = one rollout wave of --envs episodes + TD(lambda) targets + 25 critic + 25 actor Adam steps (COMATrainer.update).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))

import torch  # noqa: E402

EVAL_FIRST_EPISODE = 100_000_001      # far from any training episode (training wave w of E envs flies episodes 1 + w E ...)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024, help="training envs = episodes per update (the reference: 5)")
    ap.add_argument("--eval-envs", type=int, default=0, help="evaluation episodes (default: --envs; at least 64 are used)")
    ap.add_argument("--updates", type=int, default=60)
    ap.add_argument("--eval-every", type=int, default=10)
    ap.add_argument("--quirks", default="reference", choices=["reference", "fixed"],
                    help="reference: TD targets from the never-updated construction-time critic copy (SURVEY Q12); fixed: the synchronised target network")
    ap.add_argument("--graphs", action="store_true", help="record the round into hipGraphs (small --envs: the round is launch-bound)")
    ap.add_argument("--seed", type=int, default=0, help="torch seed (network initialisation, minibatch permutations)")
    ap.add_argument("--terrain", default="split", choices=["split", "random_field"])
    ap.add_argument("--grid", type=int, default=256, choices=[128, 256])
    ap.add_argument("--lr-scale", type=float, default=1.0, help="multiplies both learning rates (1 = the reference's 1e-5 / 1e-4)")
    ap.add_argument("--no-ig", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    from ippmarl.params import grid256_params
    from ippmarl.trainer import COMATrainer
    number = {128: 15, 256: 30}[args.grid]
    params = grid256_params(experiment__missions__n_agents=4, sensor__pixel__number_x=number, sensor__pixel__number_y=number)
    if args.lr_scale != 1.0:
        params["networks"]["actor"]["learning_rate"] *= args.lr_scale
        params["networks"]["critic"]["learning_rate"] *= args.lr_scale
    torch.manual_seed(args.seed)
    tr = COMATrainer(params, args.envs, quirks=args.quirks, terrain=args.terrain, graphs=args.graphs)
    n_eval = max(args.eval_envs or args.envs, 64)
    if n_eval == args.envs:
        ev = tr
    else:       # a second trainer object is only the evaluation batch: it flies the training actor's weights
        ev = COMATrainer(params, n_eval, quirks=args.quirks, terrain=args.terrain)
    eval_ids = torch.arange(EVAL_FIRST_EPISODE, EVAL_FIRST_EPISODE + n_eval, dtype=torch.int64)

    def evaluate():
        if ev is not tr:
            ev.actor.load_state_dict(tr.actor.state_dict())
        return ev.returns_on(eval_ids, "actor")

    t0 = time.perf_counter()
    baselines = {"random": ev.returns_on(eval_ids, "random")}
    if not args.no_ig:
        baselines["ig"] = ev.returns_on(eval_ids, "ig")
    torch.cuda.synchronize()
    baseline_s = time.perf_counter() - t0
    curve = [{"update": 0, "episodes_seen": 0, "eval": evaluate()}]
    print(json.dumps({"baselines": baselines, "update0": curve[0]}), flush=True)
    train_s, captured = 0.0, False
    for u in range(1, args.updates + 1):
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        stats = tr.rollout("train")
        stats.update(tr.update())
        torch.cuda.synchronize()
        train_s += time.perf_counter() - s0
        if args.graphs and not captured and u == 2:     # one eager round (library warm-up) before the recording
            tr.capture_graphs()
            captured = True
        if u % args.eval_every == 0 or u == args.updates:
            rec = {"update": u, "episodes_seen": u * args.envs, "eps": stats["eps"], "train_return": stats["episode_return"],
                   "critic_loss": stats["critic_loss"], "actor_loss": stats["actor_loss"], "eval": evaluate()}
            curve.append(rec)
            print(json.dumps(rec), flush=True)
    best = max(curve, key=lambda r: r["eval"]["episode_return"])
    out = {"what": "greedy-policy evaluation return of the COMA loop during training, beside the random walk and the IG planner on the same "
                   f"{n_eval} fixed episodes (ids {EVAL_FIRST_EPISODE}...)",
           "config": {"workload": f"BASELINE.json configs[2]: 4 UAVs, {args.grid}x{args.grid} grid, COMA actor + counterfactual critic, {args.envs} envs per update",
                      "envs": args.envs, "eval_episodes": n_eval, "updates": args.updates, "quirks": args.quirks, "hip_graphs": bool(args.graphs),
                      "terrain": args.terrain, "seed": args.seed, "lr_scale": args.lr_scale,
                      "transitions_per_update": args.envs * tr.T * tr.N, "adam_steps_per_update": 2 * tr.data_passes * tr.batch_number,
                      "actor_lr": params["networks"]["actor"]["learning_rate"], "critic_lr": params["networks"]["critic"]["learning_rate"]},
           "baselines": baselines, "curve": curve,
           "summary": {"random_return": baselines["random"]["episode_return"], "ig_return": baselines.get("ig", {}).get("episode_return"),
                       "untrained_return": curve[0]["eval"]["episode_return"], "final_return": curve[-1]["eval"]["episode_return"],
                       "best_return": best["eval"]["episode_return"], "best_at_update": best["update"],
                       "updates_per_s_incl_rollout": args.updates / train_s, "train_seconds": train_s, "baseline_seconds": baseline_s}}
    print(json.dumps(out["summary"]), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
