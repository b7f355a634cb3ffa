"""Optional third-party base classes for the drop-in surface.

The reference env IS a ``gym.Env`` (pioneer/envs/bullet/bullet_env.py:65, pioneer_knm_env.py:38), its episode cap is
``gym.wrappers.TimeLimit`` (pioneer/launch/pioneer_knm_train.py:27) and RLlib type-checks what ``register_env`` creators
return (``gym.Env`` / ``ray.rllib.env.VectorEnv``).  Neither gym nor ray is a dependency of this engine — nothing is
installed for it — so the façade classes derive from the real base classes WHERE THE PACKAGES EXIST and from ``object``
otherwise; the method surface is the same either way.
"""
import importlib
import importlib.util


def optional_attr(module: str, attr: str, default=object):
    """``getattr(import_module(module), attr)`` if ``module`` can be found and imported, else ``default``.
    ``attr`` may be dotted ("wrappers.TimeLimit")."""
    try:
        if importlib.util.find_spec(module.split(".")[0]) is None:
            return default
        obj = importlib.import_module(module)
        for part in attr.split("."):
            obj = getattr(obj, part)
        return obj
    except Exception:      # a broken optional install must not take the engine down
        return default


def have(module: str) -> bool:
    try:
        return importlib.util.find_spec(module) is not None
    except Exception:
        return False


GymEnv = optional_attr("gym", "Env")                       # bullet_env.py:65 `class BulletEnv(gym.Env, Generic[S])`
GymWrapper = optional_attr("gym", "Wrapper")               # base of gym.wrappers.TimeLimit
RLlibVectorEnv = optional_attr("ray.rllib.env.vector_env", "VectorEnv")      # ray.rllib.env.VectorEnv (RLlib 0.8.x .. 2.x)
HAVE_GYM = GymEnv is not object
HAVE_RLLIB = RLlibVectorEnv is not object


def to_gym_space(box):
    """The façade's Box as a real ``gym.spaces.Box`` when gym exists (RLlib's preprocessors isinstance-check it)."""
    return box.to_gym() if HAVE_GYM else box
