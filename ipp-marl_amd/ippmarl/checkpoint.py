"""Actor checkpoints in the reference's format (missions/coma_mission.py:425-451, coma_test.py:52-56).

The reference saves the WHOLE actor module with ``torch.save(actor_network, path)``, i.e. a pickle that names the class
``actor.network.ActorNetwork`` (or ``marl_framework.actor.network.ActorNetwork``).  ``load_reference_actor`` unpickles
such a file onto :class:`ippmarl.networks.ActorNetwork` (same layer names and shapes) without needing the reference on
the path; ``save_actor`` writes the same whole-module format for our class."""
from __future__ import annotations

import pickle
from typing import Dict

import torch

from .networks import ActorNetwork, CriticNetwork

_ALIASES = {
    ("actor.network", "ActorNetwork"): ActorNetwork,
    ("marl_framework.actor.network", "ActorNetwork"): ActorNetwork,
    ("critic.network", "CriticNetwork"): CriticNetwork,
    ("marl_framework.critic.network", "CriticNetwork"): CriticNetwork,
}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALIASES:
            return _ALIASES[(module, name)]
        return super().find_class(module, name)


class _PickleModule:
    """pickle-compatible module object for torch.load(pickle_module=...)."""
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    __name__ = "pickle"


def load_reference_actor(path: str, params: Dict, map_location="cpu") -> ActorNetwork:
    """Loads ``best_model.pth`` written by the reference (or by :func:`save_actor`) and returns an ActorNetwork with those
    weights.  Attributes our class does not know (the reference's ``device``, ``hidden_states`` ...) are ignored."""
    loaded = torch.load(path, map_location=map_location, pickle_module=_PickleModule, weights_only=False)
    state = loaded.state_dict() if isinstance(loaded, torch.nn.Module) else loaded
    actor = ActorNetwork(params)
    actor.load_state_dict(state)
    return actor


def save_actor(actor: ActorNetwork, path: str):
    torch.save(actor, path)
