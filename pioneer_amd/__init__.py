"""pioneer_amd — MI355X-native (gfx950 HIP) step/rollout engine for the Pioneer 6-DoF arm.

Drop-in for the hot path of xdralex/pioneer's ``pioneer.envs``: the per-env
``step()/reset()/observe()`` of ``PioneerKinematicEnv``.  All compute is in
``csrc/libpioneer_amd.so`` behind the C ABI of ``include/pioneer_amd.h``; this
package is the host-side mirror of the reference's Python interface.
"""
from .config import (EngineConfig, PioneerKinematicConfig, RenderConfig, SceneBody, SimulationConfig,  # noqa: F401
                     scene_box, scene_plane, scene_sphere)
from ._lib import PnrError, build_library, load_library  # noqa: F401

__all__ = ["EngineConfig", "PioneerKinematicConfig", "RenderConfig", "SimulationConfig", "SceneBody", "scene_box", "scene_plane",
           "scene_sphere",
           "PioneerVectorEnv", "PioneerKinematicEnv", "TimeLimit", "make_env", "make_vector_env", "PioneerRLlibVectorEnv",
           "PnrError", "build_library", "load_library"]


def __getattr__(name):
    # torch-dependent classes are imported lazily so `import pioneer_amd` stays light
    if name == "PioneerVectorEnv":
        from .vector_env import PioneerVectorEnv
        return PioneerVectorEnv
    if name in ("PioneerKinematicEnv", "TimeLimit", "make_env", "make_vector_env"):
        from . import env
        return getattr(env, name)
    if name == "PioneerRLlibVectorEnv":
        from .rllib_env import PioneerRLlibVectorEnv
        return PioneerRLlibVectorEnv
    raise AttributeError(name)
