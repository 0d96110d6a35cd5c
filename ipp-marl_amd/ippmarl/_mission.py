"""Pieces the deployment / comparison scripts share (IG_baseline, coma_test, random_baseline, lawn_mower): the two
evaluation metrics every one of them logs per step, and the json dump of a set of trials.

Metrics (reference: coma_test.py:84-97,177-196; utils/utils.py:43-76; utils/state.py:62-63): mean Shannon entropy over
the *target* cells (truth == 1) and the F1 score of the target class for the map thresholded at p > 0.5.  Both are
full-grid reductions on the device (``ippm_weighted_entropy`` with the truth as weights, ``ippm_f1_counts``)."""
from __future__ import annotations

import json
from typing import List, Sequence, Tuple

import torch

from . import _ffi


def f1_of(tp: int, fp: int, fn: int) -> float:
    return 2 * tp / (2 * tp + fp + fn) if (2 * tp + fp + fn) > 0 else 0.0


class MissionMetrics:
    """Mixin: needs ``self.mapping`` (with its engine).  ``f1_bracket`` collects, per evaluation, the F1 with every
    exactly-cancelled cell (p == 0.5 up to rounding noise, which is what classifies it in the reference) counted as
    free / as occupied — the attainable range."""

    f1_bracket: List[Tuple[float, float]]
    f1_cancelled: List[float]
    f1_counts_log: List[Tuple[Tuple[int, int, int], Tuple[int, int, int]]]   # per evaluation: (tp, fp, fn) at log-odds > +1e-5 / > -1e-5

    def _f1_counts(self, map_tensor: torch.Tensor, threshold: float = 0.0):
        env = self.mapping.engine.env
        counts = torch.zeros(1, 3, dtype=torch.int64, device=env.device)
        env.ctx.call("ippm_f1_counts", _ffi.ptr(map_tensor), env._p(env.truth), 1, float(threshold), _ffi.ptr(counts), 1, env.stream)
        return tuple(int(v) for v in counts[0].cpu())

    def _metrics(self, map_tensor: torch.Tensor = None):
        """(mean entropy over the target cells, F1 of the target class) of a device-resident log-odds map
        (default: the fused global map)."""
        env = self.mapping.engine.env
        m = env.glob[0] if map_tensor is None else map_tensor
        ent = torch.zeros(1, dtype=torch.float64, device=env.device)
        env.ctx.call("ippm_weighted_entropy", _ffi.ptr(m), env._p(env.truth), 1, _ffi.ptr(ent), 1, env.stream)
        target = int(env.truth_map[0].sum())
        # never-observed cells (exactly 0 here, exactly 0.5 in the reference) are class 0 for certain: keep them out of the
        # optimistic count below
        never = ~self.mapping.engine.observed.view_as(m) & (m == 0)
        m_b = torch.where(never, torch.full_like(m, -1.0), m).contiguous()
        tp_s, fp_s, fn_s = self._f1_counts(m_b, 1e-5)
        tp_l, fp_l, fn_l = self._f1_counts(m_b, -1e-5)
        if not hasattr(self, "f1_bracket") or self.f1_bracket is None:
            self.f1_bracket = []
        if getattr(self, "f1_cancelled", None) is None:
            self.f1_cancelled = []
        self.f1_bracket.append((f1_of(tp_s, fp_l, fn_s), f1_of(tp_l, fp_s, fn_l)))
        # the same two counts over the map as it is (never-observed cells included): integers that do not depend on rounding --
        # every cell but the exactly-cancelled ones is at least one measurement's log-odds away from 0
        if getattr(self, "f1_counts_log", None) is None:
            self.f1_counts_log = []
        self.f1_counts_log.append((self._f1_counts(m, 1e-5), self._f1_counts(m, -1e-5)))
        # cells within 1e-5 of p = 0.5 in log-odds (their class is rounding noise), as a share of the target cells
        self.f1_cancelled.append(((tp_l - tp_s) + (fp_l - fp_s)) / max(target, 1))
        return float(ent[0]) / target, f1_of(*self._f1_counts(m, 0.0))

    def _fuse_global(self):
        """K5: fuses the measurements currently held in the agents' slots into the global map."""
        env = self.mapping.engine.env
        env.ctx.call("ippm_fuse_global_reward", env._p(env.glob), env._p(env.code), env._p(env.rect), env._p(env.pos), env._p(env.ws),
                     env._p(env.sums), env._p(env.reward), 1, env.stream)

    # ---- a single map that several (virtual) platforms update in turn -----------------------------------------
    def _shared_sense(self, position: Sequence[int], correctness=None) -> None:
        """``Mapping.update_grid_map(position, shared_map)`` (mapping/mappings.py:32-78) on the map in engine slot 0:
        random_baseline.py and lawn_mower.py keep ONE map and let every platform update it directly."""
        import numpy as np
        eng, env = self.mapping.engine, self.mapping.engine.env
        env.pos[0, 0].copy_(torch.as_tensor(np.asarray(position, dtype=np.int32)))
        flips = None
        if correctness is not None:
            _, fc = eng.d.footprint(position)
            flips = self.mapping._pack_one(0, fc, 1 - np.asarray(correctness))
        stage = eng.stage[0]
        eng.stage[0] += 1
        env.sense(stage=stage, flips=flips, agent=0)

    def _shared_map(self) -> torch.Tensor:
        return self.mapping.engine.env.local[0, 0]


def save_mission_numbers(entropy_list, f1_list, trials: int, budget: int, path: str):
    """[{trial: {t: entropy}}, {trial: {t: f1}}] as the reference's scripts dump it (e.g. coma_test.py:206-222); the
    reference writes to a fixed absolute path, here the caller names the file."""
    entropy_metrics = {i: {t: float(entropy_list[i][t]) for t in range(budget + 2)} for i in range(trials)}
    f1_metrics = {i: {t: float(f1_list[i][t]) for t in range(budget + 2)} for i in range(trials)}
    with open(path, "w") as fp:
        json.dump([entropy_metrics, f1_metrics], fp)
    return entropy_metrics, f1_metrics
