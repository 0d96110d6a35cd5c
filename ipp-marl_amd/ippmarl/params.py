"""YAML -> nested dict, the reference's configuration surface (params.py:11-26, params.yaml)."""
from __future__ import annotations

import copy
import os
from typing import Dict

import yaml

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "params.yaml")


def load_params(config_file_path: str = DEFAULT_CONFIG) -> Dict:
    if not os.path.isfile(config_file_path):
        raise FileNotFoundError(f"Config file {config_file_path} does not exist!")
    with open(config_file_path, "rb") as f:
        return yaml.load(f.read(), Loader=yaml.Loader)


def default_params(**overrides) -> Dict:
    """Default params with dotted-path overrides written with '__', e.g. sensor__pixel__number_x=30."""
    p = copy.deepcopy(load_params())
    for key, value in overrides.items():
        node = p
        parts = key.split("__")
        for k in parts[:-1]:
            node = node[k]
        node[parts[-1]] = value
    return p


def grid256_params(**overrides) -> Dict:
    """BASELINE config 2/3: exactly 256 x 256 cells, footprint half-widths 15/30/45."""
    base = dict(sensor__field_of_view__angle_x=60.7, sensor__field_of_view__angle_y=60.7,
                sensor__pixel__number_x=30, sensor__pixel__number_y=30)
    base.update(overrides)
    return default_params(**base)
