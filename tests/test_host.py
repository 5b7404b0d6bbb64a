"""CPU tests (-m "not gpu") of the host side: config vs the reference's constants, the host
parity reset, the C-ABI library (loads, exports every declared symbol, validates arguments --
no compute without a GPU), spaces, sharding arithmetic, and the no-fallback rule."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g():
    import gym_acas2d_amd as g
    return g


def test_config_matches_reference_constants(g, oracle_mod):
    c = g.ACAS2DConfig()
    assert (c.max_steps, c.width, c.height, c.fps) == (1000, 1600, 1000, 100)        # settings.py:9,15-17
    assert (c.aircraft_size, c.collision_radius, c.goal_radius, c.safe_distance) == (24, 48, 144, 192)
    assert c.acc_lat_limit == pytest.approx(196.133) and c.airspeed == 200
    cc = c.to_c()
    assert cc.d_goal_max == 3408 and cc.d_dev_max == 2000 and cc.v_closing_max == 400
    assert cc.d_sep_max == pytest.approx(5886.796, abs=1e-3) and cc.d_cpa_max == pytest.approx(1886.796, abs=1e-3)
    assert cc.rw_d_goal_max == 3408 and cc.rw_d_dev_max == 704 and cc.collision_dist == 96
    # field-for-field against the oracle's independent restatement of settings.py
    oc = oracle_mod.default_config()
    for name, _ in oracle_mod.OracleConfig._fields_:
        if name != "_pad":
            assert getattr(cc, name) == getattr(oc, name), name
    assert c.obs_dim == 8 and g.ACAS2DConfig(n_traffic=8).obs_dim == 29
    lo, hi = g.ACAS2DConfig(n_traffic=2).obs_low_high()                                # environment.py:19-20
    assert lo == [0, 0, -1, 0, 0, 0, -1, -1, 0, -1, -1] and hi == [1] * 11
    with pytest.raises(ValueError):
        g.ACAS2DConfig(n_traffic=0)
    assert g.ACAS2DConfig.algorithmic_bytes_per_env_step(8, 4) == 361
    assert g.ACAS2DConfig.algorithmic_bytes_per_env_step(3, 4) == 181
    assert g.ACAS2DConfig.algorithmic_bytes_per_env_step(64, 8) == 4745


def test_parity_reset_reproduces_reference_draws(g):
    """random.seed(13) -> 2nd game is (48, 500, psi=358.1242450086868),
    traffic (1552, 48, 136.41722591475224); also the captured initial states for N = 1..64."""
    cfg = g.ACAS2DConfig()
    rng = random.Random(13)
    g.reset_parity.draw_episode(cfg, rng)
    own, trf, goal = g.reset_parity.draw_episode(cfg, rng)
    assert list(own) == [48, 500.0, 358.1242450086868, 200]
    assert list(trf[0]) == [1552, 48, 136.41722591475224, 200.0]
    assert list(goal) == [1456, 500.0]
    for N in (1, 3, 8, 64):
        fx = H.load("ref_rollout_n%d.npz" % N)
        o, t, gl = H.parity_reset_states(g.ACAS2DConfig(n_traffic=N), int(fx["seed_py"]), 1, len(fx["ep_own"]))
        assert np.array_equal(o, fx["ep_own"]) and np.array_equal(t, fx["ep_trf"]) and np.array_equal(gl, fx["ep_goal"])
    # the module-level default draws from the global `random`, like the reference
    random.seed(13)
    g.reset_parity.draw_episode(cfg)
    own2, _, _ = g.reset_parity.draw_episode(cfg)
    assert own2[2] == 358.1242450086868


def test_c_abi_library_loads_and_exports_every_declared_symbol(g):
    header = open(os.path.join(ROOT, "include", "acas2d.h")).read()
    declared = set(re.findall(r"\b(acas2d_[a-z0-9_]+)\s*\(", header))
    assert {"acas2d_step_f32", "acas2d_step_f64", "acas2d_rollout_f32", "acas2d_rollout_f64",
            "acas2d_rollout_policy_f32", "acas2d_rollout_policy_f64",
            "acas2d_reset_f32", "acas2d_reset_f64", "acas2d_last_error", "acas2d_abi_version"} <= declared
    L = g.native.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert set(g.native.EXPORTS) == declared
    assert L.acas2d_abi_version() == g.native.ABI_VERSION == 7 and L.acas2d_config_size() == C.sizeof(g.config.CConfig)
    assert int(re.search(r"#define ACAS2D_ABI_VERSION (\d+)", header).group(1)) == g.native.ABI_VERSION


def test_c_abi_argument_validation_needs_no_gpu(g):
    L = g.native.lib()
    cfg = g.ACAS2DConfig().to_c()
    st, io = g.native.CState(), g.native.CStepIO()
    assert L.acas2d_step_f32(None, C.byref(st), None, C.byref(io), 0, 0, 0, 4, 1, None) == -22
    assert b"NULL cfg" in L.acas2d_last_error()
    assert L.acas2d_step_f64(C.byref(cfg), C.byref(st), None, C.byref(io), 0, 0, 0, 4, 1, None) == -22
    assert b"state" in L.acas2d_last_error()
    dummy = (C.c_double * 4096)()
    base = C.addressof(dummy)
    full = g.native.CState(*([base] * 14))        # trace stays NULL
    assert L.acas2d_step_f64(C.byref(cfg), C.byref(full), None, C.byref(io), 0, 0, 0, 4, 1, None) == -22
    assert b"required" in L.acas2d_last_error()
    io_ok = g.native.CStepIO(*([base] * 5 + [None] * 3))
    assert L.acas2d_step_f64(C.byref(cfg), C.byref(full), None, C.byref(io_ok), 0, 0, 0, 4, 0, None) == -22
    assert b"n_traffic" in L.acas2d_last_error()
    assert L.acas2d_step_f64(C.byref(cfg), C.byref(full), None, C.byref(io_ok), 0, 0, 0, -1, 1, None) == -22
    assert L.acas2d_step_f64(C.byref(cfg), C.byref(full), None, C.byref(io_ok), 0, 0, 0, 0, 1, None) == 0   # no-op
    # state_out (double-buffered state, ABI 6): the layout contract of include/acas2d.h is checked before any launch
    E, N, AR = 4, 2, g.native.AUTO_RESET

    def gen2(**over):
        """A second generation E (per-env arrays) / E * N (traffic arrays) elements behind `full`'s, in doubles."""
        f = {n: getattr(full, n) for n, _ in g.native.CState._fields_}
        for n in ("own_x", "own_y", "own_psi", "total_reward"):
            f[n] = base + 8 * E
        f["steps"] = base + 4 * E
        f["trf_x"] = f["trf_y"] = base + 8 * E * N
        f.update(over)
        return g.native.CState(*[f[n] for n, _ in g.native.CState._fields_])

    def step(out, flags=AR):
        return L.acas2d_step_f64(C.byref(cfg), C.byref(full), None if out is None else C.byref(out), C.byref(io_ok), flags,
                                 0, 0, E, N, None)

    assert step(gen2(own_v=base + 64)) == -22 and b"must share" in L.acas2d_last_error()
    assert step(gen2(own_y=base + 8 * (E + 1))) == -22 and b"ONE element offset" in L.acas2d_last_error()
    assert step(gen2(steps=base + 8 * E)) == -22 and b"ONE element offset" in L.acas2d_last_error()
    assert step(gen2(trf_y=base + 8 * E * N + 8)) == -22 and b"ONE element offset" in L.acas2d_last_error()
    assert step(gen2(), flags=0) == -22 and b"ACAS2D_AUTO_RESET" in L.acas2d_last_error()
    near = {n: base + 8 * (E - 1) for n in ("own_x", "own_y", "own_psi", "total_reward")}
    assert step(gen2(steps=base + 4 * (E - 1), **near)) == -22 and b"overlaps" in L.acas2d_last_error()
    assert L.acas2d_reset_f32(C.byref(cfg), C.byref(st), None, None, 1, 0, 0, 4, 1, None) == -22
    assert L.acas2d_reset_f32(C.byref(cfg), C.byref(full), None, None, 1, 0, 0, 0, 1, None) == 0
    with pytest.raises(RuntimeError, match="acas2d: error -22"):
        g.native.check(L.acas2d_launch_geometry(-1, 1, 4, None, None, None, None))
    # headline config: 4 traffic per lane as one 16-byte vector, 2 lanes per env, 32 envs per wave
    assert g.native.launch_geometry(65536, 8) == {"lanes_per_env": 2, "traffic_per_lane": 4,
                                                  "block_threads": 256, "grid_blocks": 512}
    assert g.native.launch_geometry(65536, 8, 8)["traffic_per_lane"] == 4         # float64: 4 per lane (2 x 16 B)
    assert g.native.launch_geometry(4096, 3)["lanes_per_env"] == 1
    assert g.native.launch_geometry(10, 1)["lanes_per_env"] == 1
    assert g.native.launch_geometry(65536, 64)["lanes_per_env"] == 16
    assert L.acas2d_state_size() == C.sizeof(g.native.CState)
    # consecutive ("arena") layout, include/acas2d.h: pointer arithmetic only, so synthetic addresses do
    E, N, b = 2048, 8, 1 << 20

    def arena(elem=4, **over):
        f = {n: 0 for n, _ in g.native.CState._fields_}
        for k, n in enumerate(("own_x", "own_y", "own_psi", "total_reward", "steps")):
            f[n] = b + k * E * elem
        for k, n in enumerate(("own_v", "goal_x", "goal_y", "episode")):
            f[n] = 2 * b + k * E * elem
        f["trf_x"], f["trf_y"] = 3 * b, 3 * b + E * N * elem
        f["trf_psi"], f["trf_v"] = 4 * b, 4 * b + E * N * elem
        f["status"], f["trace"] = 5 * b, None
        f.update(over)
        return g.native.CState(*[f[n] for n, _ in g.native.CState._fields_])

    yes = lambda st, n=N, elem=4, e=E: L.acas2d_state_is_consecutive(C.byref(st), e, n, elem)  # noqa: E731
    assert yes(arena()) == 1
    assert yes(arena(own_y=b + 4 * E + 4)) == 0 and yes(arena(trf_v=4 * b)) == 0 and yes(arena(episode=6 * b)) == 0
    assert yes(arena(), e=E - 1) == 0 and yes(arena(), e=E - 1024) == 0      # rows are E apart for THIS env count only
    assert yes(arena(elem=8), elem=8) == 0                     # float32 only
    assert yes(arena(), n=5) == 0                              # no packed work shape for 5 traffic aircraft
    E = 2048 + 512                                             # whole multiples of eight workgroups only (1 024 envs at N = 8)
    assert yes(arena(), e=E) == 0
    E = 3072
    assert yes(arena(), e=E) == 1
    assert yes(arena(own_x=None)) == 0 and L.acas2d_state_is_consecutive(None, E, N, 4) == 0
    geo = g.native.launch_geometry(7, 200)                                          # generic walk
    assert geo["lanes_per_env"] == 64 and geo["traffic_per_lane"] == -1
    with pytest.raises(RuntimeError, match="LDS"):
        g.native.launch_geometry(16, 5000)


def test_no_cpu_fallback(g, monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback|no GPU"):
        g.ACAS2DVecEnv(4, 1)
    with pytest.raises(RuntimeError):
        g.ACAS2DVecEnv(4, 1, device="cpu")
    monkeypatch.setattr(g.native, "LIB_PATH", "/nonexistent/libacas2d_hip.so")
    monkeypatch.setattr(g.native, "_lib", None)
    with pytest.raises(g.native.NativeLibraryError, match="no CPU fallback"):
        g.native.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gym-acas2d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inl", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle-state", "").replace("oracle vectors", "") \
                    .replace("Oracle", ""), os.path.join(dirpath, f)


def test_spaces_and_sharding(g):
    b = g.Box(low=-1, high=1, shape=(1,), dtype=np.float64)
    assert b.shape == (1,) and b.contains(np.array([0.5])) and not b.contains(np.array([1.5]))
    tot = 1048576
    blocks = [g.shard_range(tot, r, 8) for r in range(8)]
    assert blocks[0] == (0, 131072) and blocks[7] == (917504, 131072)          # BASELINE configs[3]
    for world in (1, 2, 3, 7, 8):
        bl = [g.shard_range(1000, r, world) for r in range(world)]
        assert bl[0][0] == 0 and sum(c for _, c in bl) == 1000
        assert all(bl[i][0] + bl[i][1] == bl[i + 1][0] for i in range(world - 1))
        assert max(c for _, c in bl) - min(c for _, c in bl) <= 1
