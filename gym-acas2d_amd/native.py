"""ctypes binding of libacas2d_hip.so (include/acas2d.h).  There is NO fallback: if the HIP
library is missing or fails to load, importing the engine raises."""
import ctypes as C
import os
import subprocess

from .config import CConfig

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libacas2d_hip.so")
ABI_VERSION = 7

AUTO_RESET = 1


class CState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y", "trf_x", "trf_y", "trf_psi",
        "trf_v", "steps", "total_reward", "status", "episode", "trace")]


class CStepIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "actions", "obs", "reward", "done", "outcome", "term_obs", "ep_return", "ep_steps")]


class CPolicy(C.Structure):
    """struct Acas2dPolicy: the SB3 MlpPolicy actor, float32, first two weights transposed."""
    _fields_ = [(n, C.c_void_p) for n in ("w1t", "b1", "w2t", "b2", "w3", "b3")] + \
               [("hidden", C.c_int32), ("_pad", C.c_int32)]


class CActorCritic(C.Structure):
    """struct Acas2dActorCritic: actor + value net + log-std, the per-step value / log-prob outputs, the noise stream."""
    _fields_ = [("actor", CPolicy)] + [(n, C.c_void_p) for n in ("v1t", "vb1", "v2t", "vb2", "v3", "vb3", "log_std",
                                                                 "values", "logp")] + \
               [("noise_seed", C.c_uint64), ("noise_step", C.c_uint32), ("_pad", C.c_uint32)]


class CPpoUpdate(C.Structure):
    """struct Acas2dPpoUpdate: one PPO minibatch update (include/acas2d.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "actor_w1", "actor_b1", "actor_w2", "actor_b2", "actor_w3", "actor_b3", "critic_w1", "critic_b1", "critic_w2",
        "critic_b2", "critic_w3", "critic_b3", "log_std", "obs", "act", "old_logp", "adv", "ret", "idx")] + \
        [("n_rows", C.c_int32), ("obs_dim", C.c_int32)] + \
        [(n, C.c_float) for n in ("clip_range", "vf_coef", "ent_coef", "max_grad_norm", "learning_rate", "beta1", "beta2",
                                  "adam_eps")] + \
        [(n, C.c_void_p) for n in ("grad", "adam_m", "adam_v", "adam_step", "stats")]


EXPORTS = ("acas2d_abi_version", "acas2d_config_size", "acas2d_state_size", "acas2d_last_error", "acas2d_step_f32",
           "acas2d_step_f64", "acas2d_rollout_f32", "acas2d_rollout_f64", "acas2d_rollout_policy_f32",
           "acas2d_rollout_policy_f64", "acas2d_collect_f32", "acas2d_collect_f64", "acas2d_ppo_workspace_floats",
           "acas2d_ppo_update_f32", "acas2d_reset_f32", "acas2d_reset_f64", "acas2d_launch_geometry",
           "acas2d_state_is_consecutive")


class NativeLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", _CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout + r.stderr)
    if r.returncode:
        raise NativeLibraryError("building libacas2d_hip.so failed:\n" + r.stderr[-2000:])
    return LIB_PATH


_lib = None


def lib():
    """Load the library once.  torch must be imported first so that the HIP runtime the kernels
    register with (DT_NEEDED libamdhip64.so.7) is the one torch already loaded."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's libamdhip64 before ours resolves its DT_NEEDED)
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "%s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C %s`.  The ACAS2D engine has no CPU fallback." % (LIB_PATH, _CSRC))
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    missing = [n for n in EXPORTS if not hasattr(L, n)]
    if missing:
        raise NativeLibraryError("libacas2d_hip.so lacks symbols: %s" % missing)
    L.acas2d_abi_version.restype = C.c_int
    L.acas2d_config_size.restype = C.c_size_t
    L.acas2d_state_size.restype = C.c_size_t
    L.acas2d_last_error.restype = C.c_char_p
    for name in ("acas2d_step_f32", "acas2d_step_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CConfig), C.POINTER(CState), C.POINTER(CState), C.POINTER(CStepIO), C.c_uint32,
                      C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]      # state_out may be None
    for name in ("acas2d_rollout_f32", "acas2d_rollout_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CConfig), C.POINTER(CState), C.POINTER(CStepIO), C.c_int32,
                      C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    for name in ("acas2d_rollout_policy_f32", "acas2d_rollout_policy_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CConfig), C.POINTER(CState), C.POINTER(CStepIO), C.POINTER(CPolicy), C.c_void_p,
                      C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    for name in ("acas2d_collect_f32", "acas2d_collect_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CConfig), C.POINTER(CState), C.POINTER(CStepIO), C.POINTER(CActorCritic), C.c_void_p,
                      C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    L.acas2d_ppo_workspace_floats.restype = C.c_int
    L.acas2d_ppo_workspace_floats.argtypes = [C.c_int32]
    L.acas2d_ppo_update_f32.restype = C.c_int
    L.acas2d_ppo_update_f32.argtypes = [C.POINTER(CPpoUpdate), C.c_void_p]
    for name in ("acas2d_reset_f32", "acas2d_reset_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(CConfig), C.POINTER(CState), C.c_void_p, C.c_void_p, C.c_int32,
                      C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    L.acas2d_state_is_consecutive.restype = C.c_int
    L.acas2d_state_is_consecutive.argtypes = [C.POINTER(CState), C.c_int64, C.c_int32, C.c_int32]
    L.acas2d_launch_geometry.restype = C.c_int
    L.acas2d_launch_geometry.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    if L.acas2d_abi_version() != ABI_VERSION:
        raise NativeLibraryError("ABI version %d != %d" % (L.acas2d_abi_version(), ABI_VERSION))
    if L.acas2d_config_size() != C.sizeof(CConfig):
        raise NativeLibraryError("Acas2dConfig layout mismatch: %d != %d" %
                                 (L.acas2d_config_size(), C.sizeof(CConfig)))
    if L.acas2d_state_size() != C.sizeof(CState):
        raise NativeLibraryError("Acas2dState layout mismatch: %d != %d" % (L.acas2d_state_size(), C.sizeof(CState)))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RuntimeError("acas2d: error %d: %s" % (rc, lib().acas2d_last_error().decode()))


def launch_geometry(n_envs, n_traffic, elem_size=4):
    g, c, b, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    check(lib().acas2d_launch_geometry(n_envs, n_traffic, elem_size, C.byref(g), C.byref(c), C.byref(b),
                                       C.byref(n)))
    return {"lanes_per_env": g.value, "traffic_per_lane": c.value, "block_threads": b.value,
            "grid_blocks": n.value}
