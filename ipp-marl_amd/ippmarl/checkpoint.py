"""Actor checkpoints in the reference's format (missions/coma_mission.py:425-451, coma_test.py:52-56).

The reference saves the WHOLE actor module with ``torch.save(actor_network, path)``, i.e. a pickle that names the class
``actor.network.ActorNetwork`` (or ``marl_framework.actor.network.ActorNetwork``).  ``load_reference_actor`` unpickles
such a file onto :class:`ippmarl.networks.ActorNetwork` (same layer names and shapes) without needing the reference on
the path.  ``save_actor`` writes a whole-module pickle that names the REFERENCE's class path, so the reference's own
``torch.load(best_model.pth)`` (coma_test.py:52-56) resolves it to its ``actor.network.ActorNetwork`` without this package
(or libippmarl.so) being importable, and puts the plain ``state_dict`` next to it."""
from __future__ import annotations

import pickle
import threading
from typing import Dict

import torch

from .networks import ActorNetwork, CriticNetwork

_ALIASES = {
    ("actor.network", "ActorNetwork"): ActorNetwork,
    ("marl_framework.actor.network", "ActorNetwork"): ActorNetwork,
    ("critic.network", "CriticNetwork"): CriticNetwork,
    ("marl_framework.critic.network", "CriticNetwork"): CriticNetwork,
}


_SAVE_LOCK = threading.Lock()   # save_actor re-registers the class under another module path for the duration of a save


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


def save_actor(actor: ActorNetwork, path: str, reference_class_path: str = "actor.network"):
    """Whole-module pickle of the actor under the reference's class path + ``<path>.state_dict`` (weights only).

    pickle stores classes by (module, name) and insists that importing that name yields the very class being pickled, so
    for the duration of the save the class is registered under ``reference_class_path`` (the reference's scripts run with
    marl_framework/ as the working directory: ``actor.network``).  Only module state travels in the pickle (parameters,
    buffers, the ``params`` dict, plain attributes), all of which the reference's class accepts through ``__setstate__``; the
    reference's ``__init__`` does not run on load, so ActorNetwork carries the plain attributes its methods read (``device``, the
    epsilon schedule, ``log_softmax`` ...) under the reference's names."""
    import sys
    import types
    cls = type(actor)
    parts = reference_class_path.split(".")
    with _SAVE_LOCK:   # (process-global state: cls.__module__ and sys.modules)
        saved_module = cls.__module__
        created = []
        for k in range(1, len(parts) + 1):
            name = ".".join(parts[:k])
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
                created.append(name)
        holder = sys.modules[reference_class_path]
        had = getattr(holder, cls.__name__, None)
        try:
            setattr(holder, cls.__name__, cls)
            cls.__module__ = reference_class_path
            torch.save(actor, path)
        finally:
            cls.__module__ = saved_module
            if had is None:
                delattr(holder, cls.__name__)
            else:
                setattr(holder, cls.__name__, had)
            for name in created:
                del sys.modules[name]
    torch.save(actor.state_dict(), path + ".state_dict")
