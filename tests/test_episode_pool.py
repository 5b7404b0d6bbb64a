"""GPU tests (-m gpu) of the episode pool (include/acas2d.h, Acas2dState.pool): the next two episodes of
every env pre-generated in HBM, so that ACAS2DGame.__init__ + the first observe() of a finished env
(game.py:80-116, environment.py:44-48) are not computed at the end of the step launch.  The pool is a
CACHE of the reset distribution: every result must be bit-identical with and without it, in every
situation that can leave a slot stale, and in steady state (almost) every reset must be served from it."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

os.environ["ACAS2D_POOL_STATS"] = "1"        # event counters in the pool header (off by default: atomics)

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def g():
    import gym_acas2d_amd as g
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    g.native.lib()
    return g


def _snapshot(v, out):
    obs, rew, done, infos = out
    return [obs.clone(), rew.clone(), done.clone(), infos.outcome.clone(), infos.episode_return.clone(),
            infos.episode_steps.clone(), infos.terminal_observation.clone(), v.own_x.clone(), v.own_psi.clone(),
            v.trf_x.clone(), v.trf_y.clone(), v.trf_psi.clone(), v.trf_v.clone(), v.steps.clone(),
            v.total_reward.clone(), v.episode.clone()]


def _same(a, b, what):
    for k, (x, y) in enumerate(zip(a, b)):
        if x.is_floating_point():
            assert torch.equal(torch.nan_to_num(x, nan=12345.0), torch.nan_to_num(y, nan=12345.0)), (what, k)
        else:
            assert torch.equal(x, y), (what, k)


def _pair(g, E, N, dtype, seed=21, config=None):
    a = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype, seed=seed, config=config, episode_pool=True)
    b = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=dtype, seed=seed, config=config, episode_pool=False)
    assert a.pool is not None and b.pool is None
    return a, b


@pytest.mark.parametrize("dtype_name,N,E,T", (("float32", 8, 4096 + 37, 400), ("float32", 3, 5000, 300),
                                              ("float32", 1, 3000, 900), ("float32", 16, 1024, 120),
                                              ("float32", 4, 2048, 200), ("float32", 2, 2000, 300)))
def test_pool_is_a_pure_cache(g, dtype_name, N, E, T):
    dtype = getattr(torch, dtype_name)
    a, b = _pair(g, E, N, dtype)
    _same([a.reset().clone()], [b.reset().clone()], "reset")
    gen = torch.Generator(device="cuda:0").manual_seed(7)
    finished, c_early, f_early = 0, None, 0
    for t in range(T):
        act = torch.rand(E, generator=gen, device="cuda:0", dtype=dtype) * 2 - 1
        sa, sb = _snapshot(a, a.step(act)), _snapshot(b, b.step(act))
        _same(sa, sb, "step %d" % t)
        finished += int(sa[2].sum())
        if t == 4:
            c_early, f_early = a.pool_counters(), finished
    c = a.pool_counters()
    assert finished > 20 and c["pool"] + c["in_step"] == finished, (finished, c)
    # A reset is generated inside the step only when more envs of one wave may finish at once than the wave
    # prefetches first observations for (kPoolRows = 4): the burst right after reset() (a fifth of the fresh
    # episodes start inside a collision disc and end at step 1, game.py:109-110) and hardly ever after it.
    late, late_pool = finished - f_early, c["pool"] - c_early["pool"]
    assert late > 20 and late_pool >= 0.99 * late, (finished, c_early, c)
    assert c["refilled"] >= c["pool"] + c["in_step"] - E      # every reset asked for one refill (the last step's are pending)


def test_pool_with_bursts_of_simultaneous_and_back_to_back_finishes(g):
    """max_steps = 3: every env times out at the same step, every third step -- whole waves finish at once
    (far more than the 4 first observations a wave prefetches: pool commits and in-step generation side by
    side), and an env whose fresh episode starts inside a collision disc finishes in consecutive steps."""
    for dtype, N in ((torch.float32, 8), (torch.float32, 3), (torch.float32, 2)):
        cfg = g.ACAS2DConfig(n_traffic=N, max_steps=3)
        E = 2048 + 5
        a, b = _pair(g, E, N, dtype, seed=3, config=cfg)
        a.reset(); b.reset()
        gen = torch.Generator(device="cuda:0").manual_seed(1)
        total = 0
        for t in range(40):
            act = torch.rand(E, generator=gen, device="cuda:0", dtype=dtype) * 2 - 1
            sa, sb = _snapshot(a, a.step(act)), _snapshot(b, b.step(act))
            _same(sa, sb, ("burst", N, t))
            total += int(sa[2].sum())
        c = a.pool_counters()
        assert total > 10 * E and c["pool"] + c["in_step"] == total and c["pool"] > 0 and c["in_step"] > 0, (total, c)


def test_pool_survives_everything_that_can_leave_it_stale(g):
    """reset_masked(), set_state(), a fused rollout (which does not maintain the pool), a checkpoint
    restore, another seed, and a caller that edits episode[] behind the engine's back: same bits as the
    env without a pool after each of them."""
    dtype, N, E = torch.float32, 8, 3072
    a, b = _pair(g, E, N, dtype, seed=5)
    a.reset(); b.reset()
    gen = torch.Generator(device="cuda:0").manual_seed(11)

    def steps(n, what):
        for t in range(n):
            act = torch.rand(E, generator=gen, device="cuda:0", dtype=dtype) * 2 - 1
            _same(_snapshot(a, a.step(act)), _snapshot(b, b.step(act)), (what, t))

    steps(60, "fresh")
    m = (torch.arange(E, device="cuda:0") % 7 == 0)
    _same([a.reset_masked(m).clone()], [b.reset_masked(m).clone()], "reset_masked")
    steps(60, "after reset_masked")
    acts = torch.rand(50, E, generator=gen, device="cuda:0", dtype=dtype) * 2 - 1
    ra, rb = a.rollout(acts), b.rollout(acts)
    for k in ("obs", "reward", "done_u8", "outcome"):
        assert torch.equal(ra[k], rb[k]), k
    steps(60, "after rollout")
    sd = {k: v.clone() for k, v in a.state_dict().items()}
    steps(30, "before restore")
    a.load_state_dict(sd); b.load_state_dict(sd)
    steps(60, "after restore")
    a.seed(77); b.seed(77)
    steps(60, "after seed")
    before = a.pool_counters()
    a.episode.add_(5); b.episode.add_(5)                      # behind the engine's back: every slot is stale now
    steps(80, "after editing episode[]")
    after = a.pool_counters()
    assert after["in_step"] > before["in_step"]              # stale slots were ignored, not used
    own = np.tile([[48.0, 500.0, 0.0, 200.0]], (E, 1))
    trf = np.tile([[[600.0, 400.0, 180.0, 200.0]] * N], (E, 1, 1))
    a.set_state(own, trf); b.set_state(own, trf)
    steps(300, "after set_state: head-on, everybody collides in the same steps")


def test_pool_at_the_headline_size_serves_every_reset(g):
    E, N = 65536, 8
    a = g.ACAS2DVecEnv(E, N, device="cuda:0", dtype=torch.float32, seed=13)
    a.reset()
    gen = torch.Generator(device="cuda:0").manual_seed(2)
    acts = torch.rand(64, E, generator=gen, device="cuda:0") * 2 - 1
    for t in range(400):
        a.step_from(acts[t % 64])
    c0 = a.pool_counters()
    for t in range(400):
        a.step_from(acts[t % 64])
    c1 = a.pool_counters()
    served, slow = c1["pool"] - c0["pool"], c1["in_step"] - c0["in_step"]
    assert served > 50000 and slow <= 0.001 * served, (c0, c1)
