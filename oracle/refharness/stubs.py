"""Inert stand-ins for the two third-party packages the reference imports but that
contribute NO arithmetic to the step path:

  * ``gym``    -- base class, Box metadata, registry  (reference: gym_ACAS2D/__init__.py:1,
                  envs/environment.py:1-3)
  * ``pygame`` -- window, fonts, images, Clock.tick (a sleep)  (reference: envs/game.py:2,12-25)

TEST INFRASTRUCTURE ONLY.  Used in the build container to import the *unmodified*
reference from /root/reference and capture golden vectors (capture_golden.py).  Never
shipped to / needed on the GPU box; nothing in the product path imports this.
"""
import sys
import types


class _Anything:
    """Object whose every attribute is callable and returns another such object."""

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()

    def __iter__(self):
        return iter(())

    def tick(self, fps=0):      # pygame.time.Clock().tick -> no sleep
        return 0


class _Box:
    def __init__(self, low, high, shape=None, dtype=None):
        import numpy as np
        self.low, self.high, self.dtype = low, high, dtype
        self.shape = shape if shape is not None else np.shape(low)


_REGISTRY = {}


def _register(id, entry_point, **kw):
    _REGISTRY[id] = entry_point


def install():
    if "gym" in sys.modules and getattr(sys.modules["gym"], "_acas2d_stub", False):
        return _REGISTRY
    gym = types.ModuleType("gym")
    gym._acas2d_stub = True
    gym.Env = type("Env", (), {})
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = _Box
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.register = _register
    gym.spaces, gym.envs, envs.registration = spaces, envs, reg
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.envs": envs,
                        "gym.envs.registration": reg})

    pg = types.ModuleType("pygame")
    pg._acas2d_stub = True
    pg.QUIT = 256
    pg.init = lambda *a, **k: None
    for name in ("display", "image", "font", "event", "draw", "time", "transform", "key"):
        setattr(pg, name, _Anything())
    sys.modules["pygame"] = pg
    return _REGISTRY
