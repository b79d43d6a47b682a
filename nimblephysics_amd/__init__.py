"""nimblephysics_amd — MI355X-native batched differentiable timestep.

Drop-in for ONE hot path of nimblephysics: `timestep(world, state, action)`
(python/nimblephysics/timestep.py:63-69).  See DESIGN.md / INTEGRATION.md.
"""
from .model import (BodySpec, BoxSpec, CapsuleSpec, ModelDescription, SphereSpec, atlas, box_stack, cartpole,  # noqa: F401
                    make_transform, single_pendulum)

__all__ = ["ModelDescription", "BodySpec", "BoxSpec", "SphereSpec", "CapsuleSpec", "World", "timestep", "TimestepLayer", "rollout", "RolloutLayer", "single_pendulum", "cartpole",
           "atlas", "box_stack", "make_transform", "load_urdf", "load_skel", "with_ground", "load_model", "loadWorld", "model_from_nimble_world", "WrtMassBodyNodeEntryType", "GraphedStep", "GraphedRollout", "neural", "forwardPass", "BackpropSnapshot",
           "LossGradient", "LossGradientHighLevelAPI", "NimbleAmdError"]


def __getattr__(name):
    if name == "NimbleAmdError":                             # what every refused model / failed call raises
        from ._lib import NimbleAmdError
        return NimbleAmdError
    if name in ("GraphedStep", "GraphedRollout"):
        from . import graph as _g
        return getattr(_g, name)
    if name == "WrtMassBodyNodeEntryType":
        from .mass import WrtMassBodyNodeEntryType
        return WrtMassBodyNodeEntryType
    if name == "model_from_nimble_world":
        from .extract import model_from_nimble_world
        return model_from_nimble_world
    if name in ("load_urdf", "load_skel", "with_ground", "load_model", "loadWorld", "load_model", "loadWorld"):
        from . import loaders as _l
        return getattr(_l, name)
    if name in ("neural", "forwardPass", "BackpropSnapshot", "LossGradient", "LossGradientHighLevelAPI"):
        import importlib
        mod = importlib.import_module(".neural", __name__)
        return mod if name == "neural" else getattr(mod, name)
    # torch-dependent pieces are imported lazily so that model building works without a GPU stack
    if name in ("World",):
        from .world import World
        return World
    if name in ("timestep", "TimestepLayer", "rollout", "RolloutLayer"):
        from . import timestep as _t
        return getattr(_t, name)
    raise AttributeError(name)
