#!/usr/bin/env python
"""One warm-up round + one timed COMA round (rollout with the actor, TD targets, 25+25 Adam steps) for rocprofv3."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch  # noqa: E402

from ippmarl.params import grid256_params  # noqa: E402
from ippmarl.trainer import COMATrainer  # noqa: E402

tr = COMATrainer(grid256_params(), int(os.environ.get("ENVS", 1024)), terrain=os.environ.get("TERRAIN", "split"))
if os.environ.get("ROLLOUT_ONLY", "0") == "1":   # counter passes: the rollout's map kernels only (three stream copies of the local maps first:
    scratch = torch.empty_like(tr.env.local)      # the calibration of tools/pmc_summary.py)
    for _ in range(3):
        tr.env.ctx.call("ippm_stream_copy", tr.env._p(tr.env.local), tr.env._p(scratch), tr.env.local.numel() * 4, tr.env.stream)
    tr.rollout("train"); tr.filled = 0
    tr.rollout("train"); tr.filled = 0
    torch.cuda.synchronize()
    print("two rollouts")
    raise SystemExit(0)
tr.rollout("train"); tr.update()
torch.cuda.synchronize()
t0 = time.perf_counter(); tr.rollout("train"); torch.cuda.synchronize(); t1 = time.perf_counter()
tr.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"rollout {t1 - t0:.3f}s update {t2 - t1:.3f}s")
if os.environ.get("MIOPEN_FIND", "0") == "1":   # second measurement with MIOpen's exhaustive find
    torch.backends.cudnn.benchmark = True
    tr.rollout("train"); tr.update(); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.rollout("train"); torch.cuda.synchronize(); t1 = time.perf_counter()
    tr.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"cudnn.benchmark=True: rollout {t1 - t0:.3f}s update {t2 - t1:.3f}s")
if os.environ.get("CHANNELS_LAST", "0") == "1":   # conv weights in channels_last as well (inputs already are)
    for net in (tr.actor, tr.critic, tr.frozen_target, tr.critic_learner.target_critic):
        net.to(memory_format=torch.channels_last)
    tr.rollout("train"); tr.update(); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.rollout("train"); torch.cuda.synchronize(); t1 = time.perf_counter()
    tr.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"channels_last weights: rollout {t1 - t0:.3f}s update {t2 - t1:.3f}s")
