"""ACAS2DVecEnv -- E independent ACAS2D episodes resident in HBM, advanced by one HIP kernel
launch per ``step()``.

Boundary: the reference's ``ACAS2DEnv.reset()/step()`` surface
(gym_ACAS2D/envs/environment.py:29-48) widened to a batch with Stable-Baselines3 ``VecEnv``
semantics, because that is what ``training_main.py:44-52`` wraps the env in
(``Monitor`` + ``DummyVecEnv``): auto-reset on done, the returned observation is the new
episode's first observation, the finished episode's last observation and its return / length
are reported through ``infos[i]["terminal_observation"]`` / ``infos[i]["episode"]``.
(SB3 1.1.0 is not vendored in the reference: these semantics are *parity unpinned* by any
reference test and follow SB3's documented behaviour.)

All tensors live on the GPU; nothing here computes step arithmetic on the host.
"""
import ctypes as C
from collections.abc import Sequence

import numpy as np
import torch

from . import native
from .config import ACAS2DConfig
from .spaces import Box

_STATE_FIELDS = ("own_x", "own_y", "own_psi", "own_v", "goal_x", "goal_y",
                 "trf_x", "trf_y", "trf_psi", "trf_v")
# columns of ACAS2DVecEnv.trace (testing_main.py:123-137's names; game.py:132-160, :231-241, :266-276)
# the arrays a step rewrites for every env: double-buffered (include/acas2d.h, acas2d_step_*'s state_out)
_GENERATION_FIELDS = ("own_x", "own_y", "own_psi", "trf_x", "trf_y", "steps", "total_reward")
TRACE_COLUMNS = ("psi", "d_sep", "a_lat", "d_goal", "delta_heading", "v_closing", "d_cpa", "d_dev",
                 "r_d_goal", "r_h_goal", "r_d_cpa", "r_d_dev", "r_step")


class LazyInfos(Sequence):
    """``infos`` of one step as a sequence of dicts built on demand (a Python list of 65 536
    dicts per step would dominate the step time).  Tensor views of the same data:
    ``done``, ``outcome``, ``terminal_observation``, ``episode_return``, ``episode_steps``."""

    def __init__(self, env):
        self._env = env
        self.done = env._done.view(torch.bool)
        self.outcome = env._outcome
        self.terminal_observation = env._term_obs
        self.episode_return = env._ep_return
        self.episode_steps = env._ep_steps
        self._host = None

    def __len__(self):
        return self._env.num_envs

    def _fetch(self):
        if self._host is None:
            done = self.done.cpu().numpy()
            idx = np.nonzero(done)[0]
            h = {"done": done, "idx": idx}
            if len(idx) and self._env.auto_reset:
                sel = torch.as_tensor(idx, device=self.done.device)
                h["outcome"] = self.outcome[sel].cpu().numpy()
                h["r"] = self.episode_return[sel].cpu().numpy()
                h["steps"] = self.episode_steps[sel].cpu().numpy()
                h["tobs"] = (self.terminal_observation[sel].cpu().numpy()
                             if self.terminal_observation is not None else None)
                h["pos"] = {int(e): k for k, e in enumerate(idx)}
            self._host = h
        return self._host

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        h = self._fetch()
        i = int(i) % len(self)
        if not h["done"][i] or not self._env.auto_reset:
            return {}
        k = h["pos"][i]
        info = {"outcome": int(h["outcome"][k]),
                # Monitor semantics: l = number of step() calls = game.steps - 1 (game.py:197)
                "episode": {"r": float(h["r"][k]), "l": int(h["steps"][k]) - 1,
                            "steps": int(h["steps"][k])}}
        if h["tobs"] is not None:
            info["terminal_observation"] = h["tobs"][k]
        return info


class ACAS2DVecEnv:
    """Batched ACAS2D on one GPU.

    num_envs       envs of THIS shard (one process per GPU; see sharding.shard_range)
    n_traffic      traffic aircraft per env (reference default 1, settings.py:31-32)
    dtype          torch.float32 (throughput) or torch.float64 (parity with the float64 reference)
    seed           key of the counter-based reset RNG (reference RANDOM_SEED = 13, settings.py:28)
    env_offset     global index of this shard's env 0: episodes depend on (seed, global index,
                   episode counter) only, never on how the envs are sharded
    auto_reset     VecEnv semantics (True) or the single-env "latch the outcome, freeze the
                   traffic" semantics of the reference (False; game.py:243-245)
    record_trace   (auto_reset=False only) keep `trace` [E, 16]: the per-step record row behind
                   testing_main.py:114-138's CSV columns (include/acas2d.h, Acas2dState.trace;
                   TRACE_COLUMNS below), rewritten by every reset*() / set_state(observe=True) / step()
    double_buffer  keep TWO generations of the arrays a step rewrites for every env (own_x, own_y, own_psi, steps,
                   total_reward, trf_x, trf_y): step() reads the live one and writes the other, then they swap --
                   bit-identical to stepping in place.  Default (None): on where it was measured to pay -- auto_reset
                   envs whose step launch is a single generation of wavefronts (at most two per SIMD: 65 536 x 8
                   float32 -0.4 us per launch, stores that do not hit lines the launch loaded leave the L2s during
                   the kernel instead of in the write-back after it) -- and off for larger launches, where it was
                   measured slower (131 072 x 8: 9.0 against 7.3 us).  `env.own_x` etc. always
                   name the LIVE generation: fetch them again after a step() instead of keeping the tensor.
                   hipGraph users: a graph captures the generation it was captured at -- call
                   `align_generation(g)` (g = `generation` at capture time) before each replay and capture an
                   EVEN number of steps per graph, or construct with double_buffer=False.
    """

    metadata = {"render.modes": ["rgb_array", "human"]}

    def __init__(self, num_envs, n_traffic=1, device="cuda", dtype=torch.float32, seed=13,
                 env_offset=0, auto_reset=True, keep_terminal_obs=True, config=None, record_trace=False,
                 double_buffer=None):
        if config is None:
            config = ACAS2DConfig(n_traffic=n_traffic)
        self.config = config
        self.num_envs = int(num_envs)
        self.n_traffic = int(config.n_traffic)
        self.obs_dim = config.obs_dim
        self.dtype = dtype
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be torch.float32 or torch.float64")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ACAS2DVecEnv runs on an AMD GPU (device='cuda'); there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible to torch -- the ACAS2D engine has no CPU fallback")
        self._lib = native.lib()          # raises if the HIP extension is missing
        self._step_fn = self._lib.acas2d_step_f32 if dtype == torch.float32 else self._lib.acas2d_step_f64
        self._reset_fn = self._lib.acas2d_reset_f32 if dtype == torch.float32 else self._lib.acas2d_reset_f64
        self.seed_value = int(seed)
        self.env_offset = int(env_offset)
        self.auto_reset = bool(auto_reset)
        self._ccfg = config.to_c()

        E, N, D, dev = self.num_envs, self.n_traffic, self.obs_dim, self.device
        z = lambda *shape, dt=dtype: torch.zeros(*shape, dtype=dt, device=dev)  # noqa: E731
        if double_buffer is None:           # measured policy: one generation of waves (<= 2 per SIMD on the 1 024 SIMDs)
            geo = native.launch_geometry(self.num_envs, self.n_traffic, 4 if dtype == torch.float32 else 8)
            double_buffer = self.auto_reset and self.num_envs * geo["lanes_per_env"] <= 2 * 1024 * 64
        self.double_buffer = bool(double_buffer)
        if self.double_buffer and not self.auto_reset:
            raise ValueError("double_buffer needs auto_reset=True (the latching step leaves frozen traffic unwritten)")
        G = 2 if self.double_buffer else 1
        # generations of the per-step arrays ([G, E] / [G, E, N]; own_x ... total_reward are properties naming the live one)
        if dtype == torch.float32:
            # "arena" layout (include/acas2d.h, Acas2dState): the arrays a step reads are consecutive [k][E] rows of four
            # blocks, so five base pointers name every input and the step launch takes the kernel whose loads all leave
            # before its first scalar-load round trip.  Views below; a generation is a whole [5][E] / [2][E][N] block.
            mut, tmut, const, tconst = z(G, 5, E), z(G, 2, E, N), z(4, E), z(2, E, N)
            self._gen = {"own_x": mut[:, 0], "own_y": mut[:, 1], "own_psi": mut[:, 2], "total_reward": mut[:, 3],
                         "steps": mut[:, 4].view(torch.int32), "trf_x": tmut[:, 0], "trf_y": tmut[:, 1]}
            self.own_v, self.goal_x, self.goal_y = const[0], const[1], const[2]
            self.episode = const[3].view(torch.int32)  # bit pattern of the u32 counter
            self.trf_psi, self.trf_v = tconst[0], tconst[1]
        else:
            self._gen = {"own_x": z(G, E), "own_y": z(G, E), "own_psi": z(G, E), "trf_x": z(G, E, N), "trf_y": z(G, E, N),
                         "steps": z(G, E, dt=torch.int32), "total_reward": z(G, E)}
            self.own_v = z(E)
            self.goal_x, self.goal_y = z(E), z(E)
            self.trf_psi, self.trf_v = z(E, N), z(E, N)
            self.episode = z(E, dt=torch.int32)        # bit pattern of the u32 counter
        self._cur = 0
        self.status = z(E, dt=torch.uint8)
        if record_trace and auto_reset:
            raise ValueError("record_trace needs auto_reset=False (the reference's single-env semantics)")
        self.trace = z(E, len(TRACE_COLUMNS) + 3) if record_trace else None
        self._actions = z(E)
        self._obs = z(E, D)
        self._reward = z(E)
        self._done = z(E, dt=torch.uint8)
        self._outcome = z(E, dt=torch.uint8)
        self._term_obs = z(E, D) if (keep_terminal_obs and auto_reset) else None
        self._ep_return = z(E)
        self._ep_steps = z(E, dt=torch.int32)

        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        self._cstates = []
        self._db = set(self._gen) if self._gen["own_x"].shape[0] == 2 else set()      # the double-buffered arrays
        for gen in range(G):                # one Acas2dState per generation; everything else is shared
            self._cstates.append(native.CState(*[ptr(self._gen[n][gen if n in self._db else 0] if n in self._gen else getattr(self, n))
                                                 for n, _ in native.CState._fields_]))
        self._cio = native.CStepIO(ptr(self._actions), ptr(self._obs), ptr(self._reward), ptr(self._done),
                                   ptr(self._outcome), ptr(self._term_obs), ptr(self._ep_return),
                                   ptr(self._ep_steps))
        self._flags = native.AUTO_RESET if self.auto_reset else 0

        lo, hi = config.obs_low_high()
        np_dt = np.float32 if dtype == torch.float32 else np.float64
        self.observation_space = Box(low=np.array(lo, np_dt), high=np.array(hi, np_dt), dtype=np_dt)
        self.action_space = Box(low=-1, high=1, shape=(1,), dtype=np_dt)   # environment.py:27
        self._pending = None
        self._closed = False

    # ---- helpers ------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def _cstate(self):
        """Acas2dState of the LIVE generation (what reset / rollout / collect act on, in place)."""
        return self._cstates[self._cur]

    @property
    def consecutive_layout(self):
        """True when step() launches the kernel whose loads all go through preloaded base pointers (include/acas2d.h,
        "Consecutive layout"): float32 state as allocated here, a packed work shape for n_traffic, auto_reset, and
        num_envs a whole multiple of eight workgroups' envs (1 024 at n_traffic = 8; 2 048 at 1-3; 128 at 64)."""
        return bool(self.auto_reset and self._lib.acas2d_state_is_consecutive(
            C.byref(self._cstate), self.num_envs, self.n_traffic, 4 if self.dtype == torch.float32 else 8))

    @property
    def generation(self):
        """Index (0 / 1) of the live generation of the double-buffered arrays."""
        return self._cur

    def align_generation(self, g):
        """Make generation `g` the live one (copying the seven arrays across if it is not): what a captured
        hipGraph needs before a replay when other step() calls may have run since its capture."""
        g = int(g)
        if g != self._cur:
            if not self.double_buffer:
                raise RuntimeError("align_generation(%d) on an env that steps in place" % g)
            with torch.cuda.device(self.device):
                for n, t in self._gen.items():
                    if n in self._db:
                        t[g].copy_(t[self._cur])
            self._cur = g

    def set_double_buffer(self, flag):
        """Turn the double-buffered stepping off (always possible: the live generation is then stepped in place) or
        back on (only for an env constructed with it)."""
        flag = bool(flag)
        if flag and self._gen["own_x"].shape[0] != 2:
            raise RuntimeError("this env was constructed with double_buffer=False")
        self.double_buffer = flag

    def _launch_step(self, io):
        """One acas2d_step_* launch on the live generation; with double buffering it writes the other one, which
        becomes the live one."""
        if self.double_buffer:
            nxt = 1 - self._cur
            native.check(self._step_fn(C.byref(self._ccfg), C.byref(self._cstates[self._cur]), C.byref(self._cstates[nxt]),
                                       C.byref(io), self._flags, self.seed_value, self.env_offset, self.num_envs,
                                       self.n_traffic, self._stream()))
            self._cur = nxt
        else:
            native.check(self._step_fn(C.byref(self._ccfg), C.byref(self._cstates[self._cur]), None, C.byref(io),
                                       self._flags, self.seed_value, self.env_offset, self.num_envs, self.n_traffic,
                                       self._stream()))

    def _launch_reset(self, mask, do_init, with_obs=True):
        native.check(self._reset_fn(
            C.byref(self._ccfg), C.byref(self._cstate), None if mask is None else mask.data_ptr(),
            self._obs.data_ptr() if with_obs else None, int(do_init), self.seed_value,
            self.env_offset, self.num_envs, self.n_traffic, self._stream()))

    # ---- gym / VecEnv surface -------------------------------------------------------------------
    def seed(self, seed=None):
        if seed is not None:
            self.seed_value = int(seed)
        return [self.seed_value + i for i in range(min(self.num_envs, 16))]

    def reset(self):
        """environment.py:44-48 for every env: fresh episodes from the counter-based RNG
        (episode counters restart at 0), first observation returned ([E, 5+3N], device tensor)."""
        with torch.cuda.device(self.device):
            self.episode.zero_()
            self._launch_reset(None, do_init=1)
        return self._obs

    def reset_masked(self, mask):
        """Re-initialise the envs with mask != 0 (uint8/bool tensor [E]); their episode counter
        is advanced first so that they get a new episode."""
        m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            self.episode.add_(m.to(torch.int32))
            self._launch_reset(m, do_init=1)
        return self._obs

    def set_state(self, own, traffic, goal=None, steps=None, observe=True):
        """Inject state (oracle vectors / host parity reset): own [E,4] = (x, y, psi, v),
        traffic [E,N,4], goal [E,2] or [2] (default: the config's goal), steps [E] = game.steps
        BEFORE the next observe().  With observe=True (what ACAS2DEnv.reset() does,
        environment.py:47) returns the observation of that state and increments steps
        (game.py:197); with observe=False the state is left exactly as given."""
        E, N = self.num_envs, self.n_traffic
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=self.dtype).to(self.device)  # noqa: E731
        own, traffic = t(own).reshape(E, 4), t(traffic).reshape(E, N, 4)
        goal = t(self.config.goal if goal is None else goal)
        goal = goal.expand(E, 2) if goal.dim() == 1 else goal.reshape(E, 2)
        with torch.cuda.device(self.device):
            for k, name in enumerate(("own_x", "own_y", "own_psi", "own_v")):
                getattr(self, name).copy_(own[:, k])
            for k, name in enumerate(("trf_x", "trf_y", "trf_psi", "trf_v")):
                getattr(self, name).copy_(traffic[:, :, k])
            self.goal_x.copy_(goal[:, 0])
            self.goal_y.copy_(goal[:, 1])
            if steps is None:
                self.steps.zero_()
            else:
                self.steps.copy_(torch.as_tensor(np.asarray(steps), dtype=torch.int32).to(self.device).reshape(E))
            self._launch_reset(None, do_init=0, with_obs=observe)
        return self._obs if observe else None

    def step_async(self, actions):
        """game.py:225 takes action[0] in [-1, 1]; accepts [E], [E,1], numpy or tensor."""
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions, dtype=np.float64))
        a = actions.reshape(-1)
        if a.numel() != self.num_envs:
            raise ValueError("expected %d actions, got %d" % (self.num_envs, a.numel()))
        with torch.cuda.device(self.device):
            self._actions.copy_(a, non_blocking=True)
            self._launch_step(self._cio)
        self._pending = True

    def step_wait(self):
        self._pending = None
        return self._obs, self._reward, self._done.view(torch.bool), LazyInfos(self)

    def step(self, actions):
        """environment.py:29-42 for every env.  Returns (obs [E,D], reward [E], done [E] bool,
        infos) as device tensors that are overwritten by the next step()."""
        self.step_async(actions)
        return self.step_wait()

    def step_inplace(self):
        """Launch one step reading the actions already stored in ``self.actions_buffer`` --
        the zero-copy path for on-device policies and the benchmark (graph-capturable)."""
        self._launch_step(self._cio)

    def step_from(self, actions_ptr_tensor):
        """Launch one step reading actions from another resident tensor ([E], same dtype) without
        copying them (e.g. row t of a pre-generated [T, E] action buffer)."""
        a = actions_ptr_tensor
        assert a.dtype == self.dtype and a.numel() == self.num_envs and a.is_contiguous() and a.device == self.device
        io = native.CStepIO.from_buffer_copy(self._cio)
        io.actions = a.data_ptr()
        self._launch_step(io)

    def rollout(self, actions, out=None, keep_terminal_obs=False):
        """T consecutive step() calls fused into ONE kernel launch (acas2d_rollout_*): the inner loop
        of a rollout collector when the actions are known up front (scripted / random policies:
        baseline_main.py:39-61) -- state stays in registers, no kernel boundary between steps.

        actions  [T, E] (or [T, E, 1]) tensor of this env's dtype on its device.
        Returns a dict of device tensors: obs [T, E, D], reward [T, E], done [T, E] bool,
        outcome [T, E] uint8, episode_return [T, E], episode_steps [T, E] (the last two, and
        terminal_observation [T, E, D] if keep_terminal_obs, are written only where done).
        Bit-identical to T step() calls; VecEnv auto-reset semantics.  Pass the returned dict back
        as `out` to reuse the buffers."""
        if not self.auto_reset:
            raise RuntimeError("rollout() has VecEnv auto-reset semantics; construct with auto_reset=True")
        a = actions.reshape(actions.shape[0], -1)
        T, E, D = a.shape[0], self.num_envs, self.obs_dim
        if a.shape[1] != E or a.dtype != self.dtype or a.device != self.device:
            raise ValueError("actions must be a [T, %d] %s tensor on %s" % (E, self.dtype, self.device))
        a = a.contiguous()
        dev = self.device
        if out is None:
            out = {"obs": torch.empty(T, E, D, dtype=self.dtype, device=dev),
                   "reward": torch.empty(T, E, dtype=self.dtype, device=dev),
                   "done_u8": torch.empty(T, E, dtype=torch.uint8, device=dev),
                   "outcome": torch.empty(T, E, dtype=torch.uint8, device=dev),
                   "episode_return": torch.zeros(T, E, dtype=self.dtype, device=dev),
                   "episode_steps": torch.zeros(T, E, dtype=torch.int32, device=dev),
                   "terminal_observation": (torch.zeros(T, E, D, dtype=self.dtype, device=dev)
                                            if keep_terminal_obs else None)}
            out["done"] = out["done_u8"].view(torch.bool)
        assert out["obs"].shape == (T, E, D)
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        io = native.CStepIO(ptr(a), ptr(out["obs"]), ptr(out["reward"]), ptr(out["done_u8"]), ptr(out["outcome"]),
                            ptr(out.get("terminal_observation")), ptr(out["episode_return"]),
                            ptr(out["episode_steps"]))
        fn = self._lib.acas2d_rollout_f32 if self.dtype == torch.float32 else self._lib.acas2d_rollout_f64
        with torch.cuda.device(dev):
            native.check(fn(C.byref(self._ccfg), C.byref(self._cstate), C.byref(io), T, self.seed_value,
                            self.env_offset, E, self.n_traffic, self._stream()))
            self._obs.copy_(out["obs"][T - 1])        # outputs["obs"] stays "the latest observation"
        out["_actions"] = a          # keep the (possibly re-laid-out) input alive until the launch ran
        return out

    def rollout_policy(self, policy, n_steps, out=None, keep_terminal_obs=False):
        """testing_main.py:69-105 in ONE kernel launch (acas2d_rollout_policy_*): for n_steps steps,
        `action = policy.predict(obs, deterministic=True)` then `env.step(action)`, with the SB3
        MlpPolicy actor (`policy.SB3ActorPolicy` / `ppo.ActorCritic`) evaluated inside the kernel on
        the observation the previous step left -- `self.outputs["obs"]` at the start, so call
        reset() / step() / set_state(observe=True) first.  One lane per env: n_traffic in
        {1, 2, 3, 4, 8} for float32, {1, 2, 3} for float64.  Returns rollout()'s dict plus
        "actions" [T, E] (the actions taken); VecEnv auto-reset semantics."""
        if not self.auto_reset:
            raise RuntimeError("rollout_policy() has VecEnv auto-reset semantics; construct with auto_reset=True")
        T, E, D, dev = int(n_steps), self.num_envs, self.obs_dim, self.device
        w = policy.actor_weights() if hasattr(policy, "actor_weights") else policy
        w1, b1, w2, b2, w3, b3 = (torch.as_tensor(t, dtype=torch.float32).to(dev) for t in w)
        if w1.shape != (64, D) or w2.shape != (64, 64) or w3.numel() != 64:
            raise ValueError("policy must be the SB3 MlpPolicy actor %d -> 64 -> 64 -> 1, got %s %s %s"
                             % (D, tuple(w1.shape), tuple(w2.shape), tuple(w3.shape)))
        keep = [w1.t().contiguous(), b1.contiguous(), w2.t().contiguous(), b2.contiguous(),
                w3.reshape(-1).contiguous(), b3.reshape(-1).contiguous()]
        if out is None:
            out = {"obs": torch.empty(T, E, D, dtype=self.dtype, device=dev),
                   "actions": torch.empty(T, E, dtype=self.dtype, device=dev),
                   "reward": torch.empty(T, E, dtype=self.dtype, device=dev),
                   "done_u8": torch.empty(T, E, dtype=torch.uint8, device=dev),
                   "outcome": torch.empty(T, E, dtype=torch.uint8, device=dev),
                   "episode_return": torch.zeros(T, E, dtype=self.dtype, device=dev),
                   "episode_steps": torch.zeros(T, E, dtype=torch.int32, device=dev),
                   "terminal_observation": (torch.zeros(T, E, D, dtype=self.dtype, device=dev)
                                            if keep_terminal_obs else None)}
            out["done"] = out["done_u8"].view(torch.bool)
        assert out["obs"].shape == (T, E, D) and out["actions"].shape == (T, E)
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        io = native.CStepIO(ptr(out["actions"]), ptr(out["obs"]), ptr(out["reward"]), ptr(out["done_u8"]),
                            ptr(out["outcome"]), ptr(out.get("terminal_observation")), ptr(out["episode_return"]),
                            ptr(out["episode_steps"]))
        pol = native.CPolicy(*[ptr(t) for t in keep], 64, 0)
        fn = (self._lib.acas2d_rollout_policy_f32 if self.dtype == torch.float32
              else self._lib.acas2d_rollout_policy_f64)
        with torch.cuda.device(dev):
            native.check(fn(C.byref(self._ccfg), C.byref(self._cstate), C.byref(io), C.byref(pol), ptr(self._obs),
                            T, self.seed_value, self.env_offset, E, self.n_traffic, self._stream()))
            self._obs.copy_(out["obs"][T - 1])        # the observation the NEXT action would be taken on
        out["_weights"] = keep       # keep the transposed copies alive until the launch ran
        return out

    def collect(self, policy, n_steps, noise_seed=0, noise_step=0, out=None):
        """The collector of one PPO iteration in ONE kernel launch (acas2d_collect_*; SB3 `collect_rollouts` as
        training_main.py:44-52 runs it): for n_steps steps draw `a ~ N(actor(obs), exp(log_std))`, record the raw
        action, `critic(obs)` and the draw's log-probability, step the env with `clip(a, -1, 1)`.  `policy` is a
        `ppo.ActorCritic` (SB3 MlpPolicy layout).  Starts from `self.outputs["obs"]` (call reset() / step() first).
        Returns a dict of device tensors: obs [T + 1, E, D] (obs[t] is what action t was drawn on, obs[T] the
        observation the next iteration starts from), actions / values / logp / reward [T, E], done [T, E] bool,
        outcome, episode_return, episode_steps [T, E].  The noise stream depends on (noise_seed, global env index,
        noise_step + t) only.  One lane per env: n_traffic in {1, 2, 3, 4, 8} (float32), {1, 2, 3, 4} (float64)."""
        if not self.auto_reset:
            raise RuntimeError("collect() has VecEnv auto-reset semantics; construct with auto_reset=True")
        T, E, D, dev = int(n_steps), self.num_envs, self.obs_dim, self.device
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32)  # noqa: E731
        pn, vn = policy.mlp_extractor.policy_net, policy.mlp_extractor.value_net
        keep = [f32(pn[0].weight).t().contiguous(), f32(pn[0].bias).contiguous(), f32(pn[2].weight).t().contiguous(),
                f32(pn[2].bias).contiguous(), f32(policy.action_net.weight).reshape(-1).contiguous(),
                f32(policy.action_net.bias).reshape(-1).contiguous(),
                f32(vn[0].weight).t().contiguous(), f32(vn[0].bias).contiguous(), f32(vn[2].weight).t().contiguous(),
                f32(vn[2].bias).contiguous(), f32(policy.value_net.weight).reshape(-1).contiguous(),
                f32(policy.value_net.bias).reshape(-1).contiguous(), f32(policy.log_std).reshape(-1).contiguous()]
        if keep[0].shape != (D, 64) or keep[2].shape != (64, 64) or keep[4].numel() != 64:
            raise ValueError("policy must be the SB3 MlpPolicy actor-critic %d -> 64 -> 64 -> 1" % D)
        if out is None:
            z = lambda *shape, dt=self.dtype: torch.zeros(*shape, dtype=dt, device=dev)  # noqa: E731
            out = {"obs": z(T + 1, E, D), "actions": z(T, E), "values": z(T, E), "logp": z(T, E), "reward": z(T, E),
                   "done_u8": z(T, E, dt=torch.uint8), "outcome": z(T, E, dt=torch.uint8), "episode_return": z(T, E),
                   "episode_steps": z(T, E, dt=torch.int32)}
            out["done"] = out["done_u8"].view(torch.bool)
        assert out["obs"].shape == (T + 1, E, D)
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        with torch.cuda.device(dev):
            out["obs"][0].copy_(self._obs)
            io = native.CStepIO(ptr(out["actions"]), ptr(out["obs"][1:]), ptr(out["reward"]), ptr(out["done_u8"]),
                                ptr(out["outcome"]), None, ptr(out["episode_return"]), ptr(out["episode_steps"]))
            ac = native.CActorCritic(native.CPolicy(*[ptr(t) for t in keep[:6]], 64, 0), *[ptr(t) for t in keep[6:]],
                                     ptr(out["values"]), ptr(out["logp"]), int(noise_seed) & (2 ** 64 - 1),
                                     int(noise_step) & 0xFFFFFFFF, 0)
            fn = self._lib.acas2d_collect_f32 if self.dtype == torch.float32 else self._lib.acas2d_collect_f64
            native.check(fn(C.byref(self._ccfg), C.byref(self._cstate), C.byref(io), C.byref(ac), ptr(out["obs"][0]), T,
                            self.seed_value, self.env_offset, E, self.n_traffic, self._stream()))
            self._obs.copy_(out["obs"][T])            # the observation the NEXT action would be drawn on
        out["_weights"] = keep       # keep the transposed copies alive until the launch ran
        return out

    # the per-step arrays: views of the live generation
    own_x = property(lambda self: self._gen["own_x"][self._cur if "own_x" in self._db else 0])
    own_y = property(lambda self: self._gen["own_y"][self._cur if "own_y" in self._db else 0])
    own_psi = property(lambda self: self._gen["own_psi"][self._cur if "own_psi" in self._db else 0])
    trf_x = property(lambda self: self._gen["trf_x"][self._cur if "trf_x" in self._db else 0])
    trf_y = property(lambda self: self._gen["trf_y"][self._cur if "trf_y" in self._db else 0])
    steps = property(lambda self: self._gen["steps"][self._cur if "steps" in self._db else 0])
    total_reward = property(lambda self: self._gen["total_reward"][self._cur if "total_reward" in self._db else 0])

    @property
    def actions_buffer(self):
        return self._actions

    @property
    def outputs(self):
        return {"obs": self._obs, "reward": self._reward, "done": self._done.view(torch.bool),
                "outcome": self._outcome, "terminal_observation": self._term_obs,
                "episode_return": self._ep_return, "episode_steps": self._ep_steps}

    def close(self):
        self._closed = True

    def render(self, mode="rgb_array", index=0):
        """One env's frame from a host copy of its state (render.py; reference game.py:316-431), off the step
        path: "rgb_array" -> uint8 [1000, 1600, 3]; "human" -> a pygame window (RendererUnavailable without pygame)."""
        from . import render as R
        scene = R.Scene.from_env(self, index)
        if mode == "rgb_array":
            return R.rgb_array(scene)
        if mode != "human":
            raise ValueError("render mode %r" % (mode,))
        if getattr(self, "_window", None) is None:
            self._window = R.PygameWindow()
        self._window.draw(scene)
        return None

    # ---- SB3 VecEnv shims (training_main.py hands the env to SB3) -------------------------------
    def get_attr(self, attr_name, indices=None):
        n = self.num_envs if indices is None else len(list(indices))
        return [getattr(self, attr_name)] * n

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        """SB3 VecEnv.env_method: one result per selected env (the call itself acts on the batch once)."""
        n = self.num_envs if indices is None else len(list(indices))
        return [getattr(self, method_name)(*args, **kwargs)] * n

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else len(list(indices))
        return [False] * n

    # ---- checkpoint: the env state IS the SoA tensors ---------------------------
    def state_dict(self):
        names = _STATE_FIELDS + ("steps", "total_reward", "status", "episode")
        return {n: getattr(self, n).clone() for n in names}

    def load_state_dict(self, sd):
        for n, v in sd.items():
            getattr(self, n).copy_(v)

    def algorithmic_bytes_per_step(self):
        s = 4 if self.dtype == torch.float32 else 8
        return self.num_envs * ACAS2DConfig.algorithmic_bytes_per_env_step(self.n_traffic, s)
