"""Import shim: the package directory is ``gym-acas2d_amd/`` (hyphen, per the repository layout),
which is not a valid Python identifier.  ``import gym_acas2d_amd`` loads that directory as the
package ``gym_acas2d_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gym-acas2d_amd")
_spec = importlib.util.spec_from_file_location(
    "gym_acas2d_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gym_acas2d_amd"] = _mod
_spec.loader.exec_module(_mod)
