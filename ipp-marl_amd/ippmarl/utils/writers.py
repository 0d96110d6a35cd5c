"""Scalar sinks with the two SummaryWriter methods the missions use (add_scalar, add_histogram are optional).
``make_writer`` hands out torch's TensorBoard writer when the tensorboard package is installed and a JSON-lines log
otherwise, so that a training run never depends on it."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Tuple


class ScalarLog:
    """In-memory / JSON-lines stand-in: ``scalars[tag] -> [(step, value), ...]``."""

    def __init__(self, log_dir: str = None):
        self.scalars: Dict[str, List[Tuple[int, float]]] = {}
        self._fp = None
        if log_dir:
            os.makedirs(log_dir, exist_ok=True)
            self._fp = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag: str, value, step: int = 0):
        value = float(value)
        self.scalars.setdefault(tag, []).append((int(step), value))
        if self._fp:
            self._fp.write(json.dumps({"tag": tag, "step": int(step), "value": value}) + "\n")

    def flush(self):
        if self._fp:
            self._fp.flush()

    def close(self):
        if self._fp:
            self._fp.close()
            self._fp = None


def make_writer(log_dir: str):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir)
    except Exception:   # tensorboard not installed
        return ScalarLog(log_dir)
